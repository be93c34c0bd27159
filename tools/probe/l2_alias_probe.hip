// Diagnostic (not product code): does the chip-wide rate at which EVERY CU streams the same weight buffer depend on how the
// eight per-wave streams are spaced in memory (L2 channel aliasing)?  One 512-thread block per CU; wave w reads a contiguous
// region that starts at w * (per_wave + pad) kilobytes.  hipcc --offload-arch=gfx950 -O3 l2_alias_probe.hip -o l2_alias_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int D>
__global__ __launch_bounds__(512) void probe(const u32x4* __restrict__ buf, int per_wave, int pad, int rot, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // rot: block b starts its sweep rot * b pieces into the region (wraps), so the CUs of an XCD are spread over the stream
    const int start = rot ? (int)((blockIdx.x * (unsigned)rot) % (unsigned)per_wave) : 0;
    const u32x4* base = buf + (size_t)wave * (per_wave + pad) * 64 + lane;
    u32x4 ring[D];
    u32x4 acc = {0, 0, 0, 0};
    auto addr = [&](int i) -> const u32x4* {
        int p = start + i;
        p = p >= per_wave ? p - per_wave : p;
        return base + (size_t)p * 64;
    };
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = *addr(d);
    for (int i = 0; i + D <= per_wave; i += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const u32x4 v = ring[d];
            int nx = i + D + d;
            nx = nx < per_wave ? nx : per_wave - 1;
            ring[d] = *addr(nx);
            acc ^= v;
            asm volatile("" ::: "memory");
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <int D>
float run(const u32x4* buf, int per_wave, int pad, int rot, unsigned* out, int nblk, int iters) {
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 2; ++i) probe<D><<<nblk, 512>>>(buf, per_wave, pad, rot, out);
    hipEventRecord(s);
    for (int i = 0; i < iters; ++i) probe<D><<<nblk, 512>>>(buf, per_wave, pad, rot, out);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms / iters * 1e3f;
}

int main(int argc, char** argv) {
    const int per_wave = 704;                       // KB per wave: one fused-layer stream (352 stages x 2 KB)
    const size_t bytes = (size_t)8 * (per_wave + 64) * 1024;
    u32x4* buf; unsigned* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 64);
    hipMemset(buf, 1, bytes); hipMemset(out, 0, 64);
    const double mb = 8.0 * per_wave / 1024.0;
    if (argc >= 4 && !strcmp(argv[1], "loop")) {     // l2_alias_probe loop SECONDS BLOCKS: stream for a while (tools/power_parts.py)
        const double secs = atof(argv[2]);
        const int nb = atoi(argv[3]);
        hipEvent_t s, e;
        hipEventCreate(&s); hipEventCreate(&e);
        double spent = 0.0;
        long launches = 0;
        while (spent < secs) {
            hipEventRecord(s);
            for (int i = 0; i < 2000; ++i) probe<8><<<nb, 512>>>(buf, per_wave, 0, 0, out);
            hipEventRecord(e); hipEventSynchronize(e);
            float ms; hipEventElapsedTime(&ms, s, e);
            spent += ms * 1e-3; launches += 2000;
        }
        printf("loop: %ld launches of %d blocks in %.2f s = %.1f us each, %.1f GB/s per CU\n", launches, nb, spent, spent / launches * 1e6,
               mb * 1e3 / (spent / launches * 1e6));
        return 0;
    }
    for (int nb : {1, 64, 128, 225, 256})
        for (int pad : {0, 1, 2, 3, 4, 5, 8, 9, 17, 33}) {
            float a = run<8>(buf, per_wave, pad, 0, out, nb, 20), b = run<16>(buf, per_wave, pad, 0, out, nb, 20);
            printf("blocks %3d pad %2d KB | D=8 %7.1f us %6.1f GB/s/CU %6.2f TB/s | D=16 %7.1f us %6.1f GB/s/CU %6.2f TB/s\n", nb, pad, a,
                   mb * 1e3 / a, mb * nb / a, b, mb * 1e3 / b, mb * nb / b);
        }
    for (int nb : {225, 256})
        for (int rot : {1, 7, 22, 88})
            for (int pad : {0, 1}) {
                float a = run<8>(buf, per_wave, pad, rot, out, nb, 20);
                printf("blocks %3d pad %2d KB rot %3d | D=8 %7.1f us %6.1f GB/s/CU %6.2f TB/s\n", nb, pad, rot, a, mb * 1e3 / a, mb * nb / a);
            }
    return 0;
}
