// Diagnostic (not product code): the chain kernel's GEMM stage loop (gemm_probe.hip, "reads one stage ahead" form) cut into
// PHASES of 32 stages with a workgroup barrier and a block of VALU work between them, the 8-stage ring refilled across
// the gap as in the kernel.  Which part of the phase structure costs streaming rate?
//   hipcc --offload-arch=gfx950 -O3 -w phase_probe.hip -o phase_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int BARRIER, int VALU, int STORES = 0, int COLD = 0>   // COLD: all phases as straight-line code (every instruction
                                               // fetched once, like the kernel); barrier between phases (0/1), VALU instructions per wave between phases
                                               // (x64), 16-byte global stores per lane at the end of the gap
__global__ __launch_bounds__(512) void probe(const u32x4* __restrict__ buf, int n_stages, float* out, unsigned long long* tm,
                                            float* sink = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = 1.0f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<u32x4*>(buf) + (size_t)wave * n_stages * 128, 0, n_stages * 2048, 0x00020000);
    const unsigned voff = lane * 16;
    u32x4 ra[8], rb[8];
    auto ld = [&](int slot, int st) {
        st = st < n_stages ? st : n_stages - 1;
        ra[slot] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, st * 2048, 0));
        rb[slot] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff + 1024, st * 2048, 0));
    };
#pragma unroll
    for (int d = 0; d < 8; ++d) ld(d, d);
    f32x16 acc[4] = {};
    const int r = lane & 31, h = lane >> 5;
    auto frag = [&](int ks, int row) {
        return *reinterpret_cast<const u32x4*>(smem + (ks >> 2) * 8192 + row * 128 + (((2 * (ks & 3) + h) ^ ((row >> 1) & 7)) << 4));
    };
    float junk = (float)lane;
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime(), tg = 0;
#pragma unroll (COLD ? 11 : 1)
    for (int s0 = 0; s0 + 32 <= (COLD ? 352 : n_stages); s0 += 32) {
        const unsigned long long p0 = __builtin_amdgcn_s_memrealtime();
        u32x4 a0 = frag(0, r), a1 = frag(0, 32 + r);
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            const int i = ks & 7;
            const u32x4 w0 = ra[i], w1 = rb[i];
            ld(i, s0 + ks + 8);
            u32x4 n0 = a0, n1 = a1;
            if (ks + 1 < 32) { n0 = frag(ks + 1, r); n1 = frag(ks + 1, 32 + r); }
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w0), __builtin_bit_cast(bf16x8, a0), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1), __builtin_bit_cast(bf16x8, a0), acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w0), __builtin_bit_cast(bf16x8, a1), acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1), __builtin_bit_cast(bf16x8, a1), acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" : "+v"(n0), "+v"(n1) : : "memory");
            a0 = n0; a1 = n1;
        }
        tg += __builtin_amdgcn_s_memrealtime() - p0;
        if (BARRIER) __syncthreads();
#pragma unroll 1
        for (int v = 0; v < VALU; ++v) {
#pragma unroll
            for (int k = 0; k < 64; ++k) junk = __builtin_fmaf(junk, 1.0001f, 0.5f);
        }
        if (STORES) {   // an epilogue's row stores: 32 rows x 32 bytes per instruction, column-blocked like the kernel's
            float* sp = sink + ((size_t)blockIdx.x * 8 + wave) * 16384 / 4 + (s0 & 1) * 0;
#pragma unroll
            for (int k = 0; k < STORES; ++k)
                *reinterpret_cast<float4*>(sp + k * 256 + lane * 4) = make_float4(junk, junk, junk, junk);
        }
        if (BARRIER && VALU) __syncthreads();
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    float t = junk;
    for (int k = 0; k < 4; ++k) for (int q = 0; q < 16; ++q) t += acc[k][q];
    if (t == 1.2345f) out[0] = t;
    if (blockIdx.x == 0 && lane == 0) { tm[wave * 2] = t1 - t0; tm[wave * 2 + 1] = tg; }
}

template <int BARRIER, int VALU, int STORES = 0, int COLD = 0>
void run(const u32x4* const* bufs, int nbuf, int n_stages, float* out, unsigned long long* tm, int nblk, const char* name) {
    static float* sink = nullptr;
    if (!sink) hipMalloc(&sink, (size_t)256 * 8 * 16384);
    hipFuncSetAttribute((const void*)probe<BARRIER, VALU, STORES, COLD>, hipFuncAttributeMaxDynamicSharedMemorySize, 159744);
    for (int i = 0; i < nbuf + 8; ++i) probe<BARRIER, VALU, STORES, COLD><<<nblk, 512, 159744>>>(bufs[i % nbuf], n_stages, out, tm, sink);
    hipDeviceSynchronize();
    unsigned long long h[16];
    hipMemcpy(h, tm, sizeof(h), hipMemcpyDeviceToHost);
    const double bytes_phase = 8.0 * 32 * 2048;     // per CU per 32-stage phase
    const int phases = n_stages / 32;
    printf("%-34s blocks %3d: wave 0 total %6.1f us, inside the GEMM phases %6.1f us (%5.1f GB/s per CU while in a phase); wave 4: %6.1f / %6.1f us\n",
           name, nblk, h[0] / 100.0, h[1] / 100.0, bytes_phase * phases / (h[1] / 100.0) * 1e-3, h[8] / 100.0, h[9] / 100.0);
}

int main() {
    const int n_stages = 352;
    const size_t bytes = (size_t)8 * n_stages * 2048;
    const int NB = 8;
    u32x4* bufs[NB]; float* out; unsigned long long* tm;
    for (int i = 0; i < NB; ++i) { hipMalloc(&bufs[i], bytes); hipMemset(bufs[i], 0, bytes); }
    hipMalloc(&out, 64); hipMalloc(&tm, 256);
    for (int nb : {225}) {
        run<0, 0>(bufs, NB, n_stages, out, tm, nb, "continuous stream");
        run<1, 0>(bufs, NB, n_stages, out, tm, nb, "barrier every 32 stages");
        run<0, 20>(bufs, NB, n_stages, out, tm, nb, "1280 VALU between phases");
        run<1, 20>(bufs, NB, n_stages, out, tm, nb, "barrier + 1280 VALU + barrier");
        run<1, 40>(bufs, NB, n_stages, out, tm, nb, "barrier + 2560 VALU + barrier");
        run<1, 20, 16>(bufs, NB, n_stages, out, tm, nb, "... 1280 VALU + 16 KB stores/wave");
        run<1, 20, 0, 1>(bufs, NB, n_stages, out, tm, nb, "barrier + 1280 VALU, straight-line");
        run<0, 0, 0, 1>(bufs, NB, n_stages, out, tm, nb, "continuous, straight-line");
    }
    return 0;
}
