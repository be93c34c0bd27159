// Diagnostic (not product code): how fast can ONE CU stream a buffer that every workgroup reads, by wave count, loads in
// flight per wave and address pattern?  hipcc --offload-arch=gfx950 -O3 stream_probe.hip -o stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int D, int MODE>   // MODE 0: each wave a contiguous region; 1: waves interleaved at 1 KB; 2: at 2 KB
__global__ void probe(const u32x4* __restrict__ buf, size_t bytes_per_block, unsigned* out, int nwaves) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t pieces = bytes_per_block / 1024;           // 1-KB pieces the block reads in total
    const size_t per_wave = pieces / nwaves;
    u32x4 ring[D];
    u32x4 acc = {0, 0, 0, 0};
    auto addr = [&](size_t i) -> const u32x4* {             // i-th piece of this wave
        size_t p;
        if (MODE == 0) p = (size_t)wave * per_wave + i;
        else if (MODE == 1) p = i * nwaves + wave;
        else p = (i >> 1) * (2 * nwaves) + 2 * wave + (i & 1);
        return buf + p * 64 + lane;
    };
#pragma unroll
    for (int d = 0; d < D; ++d) ring[d] = *addr(d);
    for (size_t i = 0; i + D <= per_wave; i += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const u32x4 v = ring[d];
            size_t nx = i + D + d;
            nx = nx < per_wave ? nx : per_wave - 1;
            ring[d] = *addr(nx);
            acc ^= v;
            asm volatile("" ::: "memory");
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;   // keep the loads alive
}

template <int D, int MODE>
float run(const u32x4* buf, size_t bytes, unsigned* out, int nblk, int nwaves, int iters) {
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 2; ++i) probe<D, MODE><<<nblk, nwaves * 64>>>(buf, bytes, out, nwaves);
    hipEventRecord(s);
    for (int i = 0; i < iters; ++i) probe<D, MODE><<<nblk, nwaves * 64>>>(buf, bytes, out, nwaves);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms / iters * 1e3f;
}

// Same kernel, but every launch streams a DIFFERENT buffer of a set larger than all L2s together (as the sampler does:
// 8 layers x 5.5 MB per step): is the per-CU rate set by L2 hits or by the fill from the Infinity Cache?
template <int D>
float run_cold(u32x4* const* bufs, int nbuf, size_t bytes, unsigned* out, int nblk, int nwaves, int iters) {
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < nbuf; ++i) probe<D, 0><<<nblk, nwaves * 64>>>(bufs[i], bytes, out, nwaves);
    hipEventRecord(s);
    for (int i = 0; i < iters; ++i) probe<D, 0><<<nblk, nwaves * 64>>>(bufs[i % nbuf], bytes, out, nwaves);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms / iters * 1e3f;
}

int main() {
    const size_t bytes = 4608 * 1024;   // one chain-B weight stream
    u32x4* buf; unsigned* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 64);
    hipMemset(buf, 1, bytes); hipMemset(out, 0, 64);
    {
        const int NB = 16;
        u32x4* bufs[NB];
        for (int i = 0; i < NB; ++i) { hipMalloc(&bufs[i], bytes); hipMemset(bufs[i], 1 + i, bytes); }
        for (int nb : {28, 225}) {
            float w = run<8, 0>(bufs[0], bytes, out, nb, 8, 32), c = run_cold<8>(bufs, NB, bytes, out, nb, 8, 32),
                  c16 = run_cold<16>(bufs, NB, bytes, out, nb, 8, 32);
            printf("blocks %3d waves 8 contiguous: same buffer every launch D=8 %6.1f | 16 buffers in turn (74 MB) D=8 %6.1f D=16 %6.1f GB/s per CU\n",
                   nb, bytes / w * 1e-3, bytes / c * 1e-3, bytes / c16 * 1e-3);
        }
    }
    const int blks[] = {1, 32, 225};
    for (int nb : blks)
        for (int nw : {4, 8, 16}) {
            float a = run<4, 0>(buf, bytes, out, nb, nw, 20), b = run<8, 0>(buf, bytes, out, nb, nw, 20),
                  c = run<16, 0>(buf, bytes, out, nb, nw, 20), d = run<8, 1>(buf, bytes, out, nb, nw, 20),
                  f = run<8, 2>(buf, bytes, out, nb, nw, 20), g = run<16, 1>(buf, bytes, out, nb, nw, 20);
            auto gb = [&](float us) { return bytes / us * 1e-3; };
            printf("blocks %3d waves %2d | contiguous D=4 %6.1f  D=8 %6.1f  D=16 %6.1f | interleaved-1K D=8 %6.1f D=16 %6.1f | interleaved-2K D=8 %6.1f  GB/s per CU\n",
                   nb, nw, gb(a), gb(b), gb(c), gb(d), gb(g), gb(f));
        }
    return 0;
}
