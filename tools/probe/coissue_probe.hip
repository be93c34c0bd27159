// Diagnostic (not product code): does a SIMD run one wave's MFMAs under another wave's VALU work?  512-thread blocks (two waves
// per SIMD); waves 0-3 loop over MFMAs (32x32x16 or 16x16x32 bf16, four independent accumulator sets), waves 4-7 loop over
// softmax-like VALU work (v_pk_fma_f32 + v_exp_f32 + v_pk_add_f32); each alone, then both.  Overlap: both ~ max; none: both ~ sum.
// hipcc --offload-arch=gfx950 -O3 coissue_probe.hip -o coissue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

template <int SHAPE, int VAR>   // SHAPE 32: 32x32x16, 16: 16x16x32; VAR bit 2: no transcendental, bit 3: dependent MFMA chain
__global__ __launch_bounds__(512) void probe(int mode, int mi, int vi, float* out) {
    // mode bit 0: MFMA waves, bit 1: VALU waves, bit 2: VALU work without the transcendental (v_pk_fma_f32 / v_pk_add_f32 only),
    // bit 3: the MFMAs form ONE dependent chain (the pipe idles between them for the result latency)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        if (!(mode & 1)) return;
        bf16x8_t a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane - j)); }
        if (SHAPE == 32) {
            f32x16_t c[4] = {};
            for (int i = 0; i < mi; ++i) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    constexpr int dummy0 = 0; const int q = (VAR & 8) ? dummy0 : (k & 3);
                    c[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[q], 0, 0, 0);
                }
            }
            out[blockIdx.x * 512 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
        } else {
            f32x4_t c[8] = {};
            for (int i = 0; i < mi; ++i) {
#pragma unroll
                for (int k = 0; k < 32; ++k) {     // same FLOPs as 16 of the 32x32x16
                    constexpr int dummy1 = 0; const int q = (VAR & 8) ? dummy1 : (k & 7);
                    c[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[q], 0, 0, 0);
                }
            }
            out[blockIdx.x * 512 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3] + c[4][0] + c[5][1] + c[6][2] + c[7][3];
        }
    } else {
        if (!(mode & 2)) return;
        f32x2_t x[16], sum = {0.f, 0.f};
        for (int j = 0; j < 16; ++j) x[j] = f32x2_t{0.01f * (lane + j), -0.02f * j};
        const f32x2_t l2 = {1.44f, 1.44f}, nm = {-3.0f, -3.0f};
        for (int i = 0; i < vi; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                f32x2_t y = __builtin_elementwise_fma(x[j], l2, nm);
                if (VAR & 16) {                 // unpacked: four v_fma_f32 (asm: -O3 would re-pack adjacent scalar FMAs)
                    float y0 = y[0], y1 = y[1];
                    asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %0, %0, %3, %2\n\tv_fma_f32 %1, %1, %3, %2"
                                 : "+v"(y0), "+v"(y1) : "v"(l2[0]), "v"(nm[0]));
                    y[0] = y0; y[1] = y1;
                } else if (!(VAR & 4)) {
                    y[0] = __builtin_amdgcn_exp2f(y[0]);
                    y[1] = __builtin_amdgcn_exp2f(y[1]);
                } else {
                    y = __builtin_elementwise_fma(y, l2, nm);
                    y = __builtin_elementwise_fma(y, nm, l2);
                }
                sum += y;
                x[j] = y;
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = sum[0] + sum[1];
    }
}

template <int SHAPE, int VAR>
float run(int mode, int mi, int vi, float* out, int nb) {
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    for (int i = 0; i < 2; ++i) probe<SHAPE, VAR><<<nb, 512>>>(mode, mi, vi, out);
    (void)hipEventRecord(s);
    for (int i = 0; i < 10; ++i) probe<SHAPE, VAR><<<nb, 512>>>(mode, mi, vi, out);
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    return ms / 10 * 1e3f;
}

// the same question inside ONE wave: every wave issues NV independent v_pk_fma_f32 behind each MFMA (all eight waves do this)
template <int SHAPE, int NV, bool UNPACKED = false>
__global__ __launch_bounds__(512) void probe_same(int mi, float* out, int with_mfma) {
    const int lane = threadIdx.x & 63;
    bf16x8_t a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane - j)); }
    f32x2_t x[8];
    for (int j = 0; j < 8; ++j) x[j] = f32x2_t{0.01f * (lane + j), -0.02f * j};
    const f32x2_t l2 = {1.0001f, 0.9999f}, nm = {-1e-6f, 1e-6f};
    float r = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16_t c[4] = {};
        for (int i = 0; i < mi; ++i) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (with_mfma) c[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[k & 3], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    if (UNPACKED) {             // one v_fma_f32 (half the arithmetic of the packed one, the same issue slot count)
                        float t = x[(k + v) & 7][0];
                        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t) : "v"(l2[0]), "v"(nm[0]));
                        x[(k + v) & 7][0] = t;
                    } else {
                        x[(k + v) & 7] = __builtin_elementwise_fma(x[(k + v) & 7], l2, nm);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        r = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    } else {
        f32x4_t c[8] = {};
        for (int i = 0; i < mi; ++i) {
#pragma unroll
            for (int k = 0; k < 32; ++k) {       // two 16x16x32 = the FLOPs of one 32x32x16: NV / 2 packed FMAs behind each
                if (with_mfma) c[k & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[k & 7], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < NV / 2; ++v) x[(k + v) & 7] = __builtin_elementwise_fma(x[(k + v) & 7], l2, nm);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        r = c[0][0] + c[1][1] + c[2][2] + c[3][3] + c[4][0] + c[5][1] + c[6][2] + c[7][3];
    }
    for (int j = 0; j < 8; ++j) r += x[j][0] + x[j][1];
    out[blockIdx.x * 512 + threadIdx.x] = r;
}
template <int SHAPE, int NV, bool UNPACKED = false>
float run_same(int mi, float* out, int nb, int with_mfma) {
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    for (int i = 0; i < 2; ++i) probe_same<SHAPE, NV, UNPACKED><<<nb, 512>>>(mi, out, with_mfma);
    (void)hipEventRecord(s);
    for (int i = 0; i < 10; ++i) probe_same<SHAPE, NV, UNPACKED><<<nb, 512>>>(mi, out, with_mfma);
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    return ms / 10 * 1e3f;
}

// Round 5 (VERDICT r4 "what's weak" #5): in `probe` the MFMA waves issue back to back, so one MFMA is always WAITING at issue for
// the matrix pipe.  If a waiting MFMA holds the SIMD's vector issue port, "sum, not max" would mean "a wave that camps on the
// port blocks its neighbour", not "waves never overlap".  Here the MFMA waves pad every MFMA with PAD x `s_nop 7` (scalar: no
// vector issue slot), so that with enough padding no MFMA ever waits at issue; PRIO raises the VALU waves' priority.
template <int SHAPE, int PAD, int PRIO>
__global__ __launch_bounds__(512) void probe_pad(int mode, int mi, int vi, float* out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        if (!(mode & 1)) return;
        bf16x8_t a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(0.001f * (lane + j)); b[j] = (__bf16)(0.002f * (lane - j)); }
        if (SHAPE == 32) {
            f32x16_t c[4] = {};
            for (int i = 0; i < mi; ++i) {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    c[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[k & 3], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < PAD; ++q) asm volatile("s_nop 7");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            out[blockIdx.x * 512 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
        } else {
            f32x4_t c[8] = {};
            for (int i = 0; i < mi; ++i) {
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    c[k & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[k & 7], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < PAD; ++q) asm volatile("s_nop 7");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            out[blockIdx.x * 512 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3] + c[4][0] + c[5][1] + c[6][2] + c[7][3];
        }
    } else {
        if (!(mode & 2)) return;
        if (PRIO) __builtin_amdgcn_s_setprio(PRIO);
        float x[16], sum = 0.f;
        for (int j = 0; j < 16; ++j) x[j] = 0.01f * (lane + j);
        const float l2 = 1.0001f, nm = -1e-6f;
        for (int i = 0; i < vi; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {         // six unpacked single-issue instructions per element (asm: -O3 would re-pack)
                float y = x[j];
                asm volatile("v_fma_f32 %0, %0, %1, %2\n\tv_fma_f32 %0, %0, %2, %1\n\tv_fma_f32 %0, %0, %1, %2\n\t"
                             "v_fma_f32 %0, %0, %2, %1\n\tv_fma_f32 %0, %0, %1, %2\n\tv_add_f32 %0, %0, %1"
                             : "+v"(y) : "v"(l2), "v"(nm));
                sum += y;
                x[j] = y;
            }
        }
        out[blockIdx.x * 512 + threadIdx.x] = sum;
    }
}
template <int SHAPE, int PAD, int PRIO>
float run_pad(int mode, int mi, int vi, float* out, int nb) {
    hipEvent_t s, e;
    (void)hipEventCreate(&s); (void)hipEventCreate(&e);
    for (int i = 0; i < 2; ++i) probe_pad<SHAPE, PAD, PRIO><<<nb, 512>>>(mode, mi, vi, out);
    (void)hipEventRecord(s);
    for (int i = 0; i < 10; ++i) probe_pad<SHAPE, PAD, PRIO><<<nb, 512>>>(mode, mi, vi, out);
    (void)hipEventRecord(e); (void)hipEventSynchronize(e);
    float ms; (void)hipEventElapsedTime(&ms, s, e);
    return ms / 10 * 1e3f;
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    const int mi = 2000, vi = 1400;
    for (int nb : {1, 256}) {
#define ROW(VAR_, what)                                                                                                        \
        printf("%3d block(s), %-38s 32x32x16: MFMA %7.1f  VALU %7.1f  both %7.1f us | 16x16x32: MFMA %7.1f  VALU %7.1f  both %7.1f us\n", nb, what,  \
               run<32, VAR_>(1, mi, vi, out, nb), run<32, VAR_>(2, mi, vi, out, nb), run<32, VAR_>(3, mi, vi, out, nb),                      \
               run<16, VAR_>(1, mi, vi, out, nb), run<16, VAR_>(2, mi, vi, out, nb), run<16, VAR_>(3, mi, vi, out, nb));
        ROW(0, "independent MFMAs, VALU with v_exp")
        ROW(4, "independent MFMAs, v_pk_fma only")
        ROW(8, "dependent MFMA chain, VALU with v_exp")
        ROW(12, "dependent MFMA chain, v_pk_fma only")
        ROW(16, "independent MFMAs, UNPACKED v_fma_f32")
#undef ROW
    }
    // same wave: NV packed FMAs behind every 32x32x16 MFMA (two waves per SIMD, all alike); "VALU only" = the same loop without the MFMAs
    for (int nb : {1, 256}) {
#define SROW(NV_)                                                                                                              \
        printf("%3d block(s), same wave, %d v_pk_fma behind every 32x32x16 MFMA: with the MFMAs %7.1f us, VALU only %7.1f us\n", nb, NV_, \
               run_same<32, NV_>(1000, out, nb, 1), run_same<32, NV_>(1000, out, nb, 0));
        SROW(0) SROW(2) SROW(4) SROW(6)
#undef SROW
#define UROW(NV_)                                                                                                              \
        printf("%3d block(s), same wave, %d UNPACKED v_fma_f32 behind every 32x32x16 MFMA: with the MFMAs %7.1f us, VALU only %7.1f us\n", nb, NV_, \
               run_same<32, NV_, true>(1000, out, nb, 1), run_same<32, NV_, true>(1000, out, nb, 0));
        UROW(2) UROW(4) UROW(6) UROW(8)
#undef UROW
    }
    // padded MFMA waves (no MFMA waits at issue once the pad covers the pipe time) against unpacked-VALU waves, one block
    {
        const int nb = 1, mi2 = 1000, vi2 = 1400;
#define PROW(SH_, PAD_, PRIO_)                                                                                                 \
        printf("padded probe, %dx%d MFMA + %d x s_nop 7, VALU waves at priority %d: MFMA %7.1f  VALU %7.1f  both %7.1f us\n", SH_, SH_, PAD_, \
               PRIO_, run_pad<SH_, PAD_, PRIO_>(1, mi2, vi2, out, nb), run_pad<SH_, PAD_, PRIO_>(2, mi2, vi2, out, nb),        \
               run_pad<SH_, PAD_, PRIO_>(3, mi2, vi2, out, nb));
        PROW(32, 0, 0) PROW(32, 1, 0) PROW(32, 2, 0) PROW(32, 3, 0) PROW(32, 4, 0) PROW(32, 6, 0)
        PROW(32, 0, 3) PROW(32, 2, 3) PROW(32, 3, 3)
        PROW(16, 0, 0) PROW(16, 1, 0) PROW(16, 2, 0) PROW(16, 3, 0)
        PROW(16, 0, 3) PROW(16, 1, 3)
#undef PROW
    }
    return 0;
}
