// Diagnostic (not product code): the chain kernel's GEMM stage (2 x 1-KB weight loads into a register ring of 8
// stages, 2 ds_read_b128 activation fragments, 4 v_mfma_f32_32x32x16_bf16) in isolation: which ingredient sets the
// per-CU streaming rate?  8 waves per block, one block per CU, every block streams the same 4.6 MB.
//   hipcc --offload-arch=gfx950 -O3 -w gemm_probe.hip -o gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MFMA, int LDS, int PREF>   // MFMAs per stage (0 / 4), LDS fragment reads (0 / 1), fragments one stage ahead
__global__ __launch_bounds__(512) void probe(const u32x4* __restrict__ buf, int n_stages, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = 1.0f;
    __syncthreads();
    const u32x4* p = buf + (size_t)wave * n_stages * 128 + lane;
    u32x4 ra[8], rb[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) { ra[d] = p[d * 128]; rb[d] = p[d * 128 + 64]; }
    f32x16 acc[4] = {};
    const int r = lane & 31, h = lane >> 5;
    auto frag = [&](int ks, int row) {
        return *reinterpret_cast<const u32x4*>(smem + (ks >> 2) * 8192 + row * 128 + (((2 * (ks & 3) + h) ^ ((row >> 1) & 7)) << 4));
    };
    u32x4 a0 = frag(0, r), a1 = frag(0, 32 + r);
    for (int s0 = 0; s0 + 32 <= n_stages; s0 += 32) {
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
            const int i = ks & 7;
            const u32x4 w0 = ra[i], w1 = rb[i];
            int nx = s0 + ks + 8;
            nx = nx < n_stages ? nx : n_stages - 1;
            ra[i] = p[nx * 128];
            rb[i] = p[nx * 128 + 64];
            u32x4 n0 = a0, n1 = a1;
            if (LDS) {
                if (PREF) { n0 = frag((ks + 1) & 31, r); n1 = frag((ks + 1) & 31, 32 + r); }
                else { a0 = frag(ks, r); a1 = frag(ks, 32 + r); }
            }
            if (PREF) __builtin_amdgcn_sched_barrier(0);
            if (MFMA) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w0), __builtin_bit_cast(bf16x8, a0), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1), __builtin_bit_cast(bf16x8, a0), acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w0), __builtin_bit_cast(bf16x8, a1), acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w1), __builtin_bit_cast(bf16x8, a1), acc[3], 0, 0, 0);
            } else {
                acc[0][0] += __builtin_bit_cast(float, w0.x ^ w1.y ^ a0.x ^ a1.y);
            }
            if (PREF) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("" : "+v"(n0), "+v"(n1) : : "memory");
                a0 = n0; a1 = n1;
            } else {
                asm volatile("" ::: "memory");
            }
        }
    }
    float t = 0;
    for (int k = 0; k < 4; ++k) for (int q = 0; q < 16; ++q) t += acc[k][q];
    if (t == 1.2345f) out[0] = t;
}

template <int MFMA, int LDS, int PREF>
float run(const u32x4* const* bufs, int nbuf, int n_stages, float* out, int nblk) {
    hipFuncSetAttribute((const void*)probe<MFMA, LDS, PREF>, hipFuncAttributeMaxDynamicSharedMemorySize, 159744);
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < nbuf; ++i) probe<MFMA, LDS, PREF><<<nblk, 512, 159744>>>(bufs[i], n_stages, out);
    hipEventRecord(s);
    const int iters = 32;
    for (int i = 0; i < iters; ++i) probe<MFMA, LDS, PREF><<<nblk, 512, 159744>>>(bufs[i % nbuf], n_stages, out);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms / iters * 1e3f;
}

int main() {
    const int n_stages = 288;                       // 8 waves x 288 x 2 KB = 4.6 MB
    const size_t bytes = (size_t)8 * n_stages * 2048;
    const int NB = 16;
    u32x4* bufs[NB]; float* out;
    for (int i = 0; i < NB; ++i) { hipMalloc(&bufs[i], bytes); hipMemset(bufs[i], 0, bytes); }
    hipMalloc(&out, 64);
    for (int nb : {1, 225}) {
        auto gb = [&](float us) { return bytes / us * 1e-3; };
        float a = run<0, 0, 0>(bufs, NB, n_stages, out, nb), b = run<4, 0, 0>(bufs, NB, n_stages, out, nb),
              c = run<0, 1, 0>(bufs, NB, n_stages, out, nb), d = run<4, 1, 0>(bufs, NB, n_stages, out, nb),
              f = run<4, 1, 1>(bufs, NB, n_stages, out, nb);
        printf("blocks %3d: loads only %6.1f | + 4 MFMA %6.1f | + 2 LDS reads %6.1f | + both (read, wait, use) %6.1f | + both, reads one stage ahead %6.1f GB/s per CU  (%.1f us)\n",
               nb, gb(a), gb(b), gb(c), gb(d), gb(f), f);
    }
    return 0;
}
