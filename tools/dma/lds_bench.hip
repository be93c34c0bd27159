// LDS read-pattern probe (diagnostic): cycles per ds_read_b128 wave-instruction for the fragment address patterns used
// by the GEMM kernels, 8 waves per CU, all CUs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ int addr_of(int pat, int lane, int it) {
    const int r = lane & 31, h = lane >> 5;
    const int c = (2 * (it & 3) + h);
    switch (pat) {
        case 0: return lane * 16;                                                        // linear: conflict-free reference
        case 1: return (r * 1040 + (c + 8 * (it & 7)) * 16) % 65536;                     // frag_a: padded 1040-B rows
        case 2: { int row = (it & 1) * 32 + r; return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }   // tile_off swizzle (row>>1)&7
        case 3: { int row = (it & 1) * 32 + r; return row * 128 + ((c ^ (row & 7)) << 4); }          // swizzle row&7
        case 4: { int row = (it & 1) * 32 + r; return row * 128 + (c << 4); }                        // no swizzle
        case 5: return (r * 272 + (c + 8 * (it & 1)) * 16);                              // H1: padded 272-B rows
        case 6: return (r * 1024 + (((c + 8 * (it & 7)) ^ (r & 15)) * 16)) % 65536;      // unpadded 1 KB rows, xor r&15
        default: return 0;
    }
}

__global__ __launch_bounds__(512) void lds_kernel(int pat, int iters, unsigned long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += 512) reinterpret_cast<float*>(smem)[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    u32x4 acc = {0, 0, 0, 0};
    int ad[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) ad[u] = addr_of(pat, lane, u);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 8) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const u32x4*>(smem + ad[u]);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
    if (acc[0] == 0x12345678u) sink[0] = 1.0f;
}

int main() {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 8 * 8); hipMalloc(&sink, 64);
    const int iters = 4096;
    const char* names[] = {"linear (reference)", "frag_a 1040-B rows", "tile_off (row>>1)&7", "swizzle row&7", "no swizzle", "H1 272-B rows", "1KB rows xor r&15"};
    for (int pat = 0; pat < 7; ++pat) {
        hipLaunchKernelGGL(lds_kernel, dim3(256), dim3(512), 0, 0, pat, iters, out, sink);
        hipDeviceSynchronize();
        unsigned long long h[256 * 8];
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < 256 * 8; ++i) s += (double)h[i];
        printf("%-22s: %.1f cycles per ds_read_b128 per wave (8 waves/CU) -> %.1f cycles per CU per instr\n", names[pat], s / (256 * 8) / iters, s / (256 * 8) / iters / 8);
    }
    return 0;
}
