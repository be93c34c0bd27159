// Diagnostic: cycles per MFMA (one wave per SIMD, 4 independent accumulators) of the bf16 shapes a split-bf16 ("bf16x3") policy could use.
// hipcc --offload-arch=gfx950 -O3 mfma_rate_probe.hip -o mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
template <int WHICH>
__global__ __launch_bounds__(256) void k(int n, float* out, unsigned long long* cyc) {
    const int lane = threadIdx.x & 63;
    f32x16_t c[4] = {};
    bf16x8_t a8, b8; s16x4_t a4, b4;
    for (int j = 0; j < 8; ++j) { a8[j] = (__bf16)(0.001f * (lane + j)); b8[j] = (__bf16)(0.002f * (lane - j)); }
    for (int j = 0; j < 4; ++j) { a4[j] = (short)(0x3f80 + lane + j); b4[j] = (short)(0x3f00 + lane - j); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if (WHICH == 0) c[q & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, c[q & 3], 0, 0, 0);
            else if (WHICH == 1) c[q & 3] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, c[q & 3], 0, 0, 0);
            else c[q & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(0.5f * lane, 0.25f * q, c[q & 3], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[WHICH] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 4); (void)hipMalloc(&cyc, 64);
    const int n = 2000;
    k<0><<<1, 256>>>(n, out, cyc); k<1><<<1, 256>>>(n, out, cyc); k<2><<<1, 256>>>(n, out, cyc);
    unsigned long long h[3]; (void)hipMemcpy(h, cyc, 24, hipMemcpyDeviceToHost);
    const char* nm[3] = {"32x32x16_bf16", "32x32x8_bf16_1k", "32x32x2_f32"};
    for (int i = 0; i < 3; ++i) printf("%-18s %.1f cycles per MFMA\n", nm[i], (double)h[i] / (n * 16.0));
    return 0;
}
