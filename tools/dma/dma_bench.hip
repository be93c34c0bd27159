// Synthetic L2 -> LDS DMA throughput probe (diagnostic only): every workgroup (512 threads, 1 per CU) streams the same
// `wbytes` weight buffer (L2-resident) through an LDS ring of DEPTH slots of TILE bytes with global_load_lds_dwordx4,
// waiting for the oldest slot with a counted vmcnt and a barrier per tile, and reads one dword per lane from each
// landed slot (so the data is consumed).  Reports GB/s per CU for each (TILE, DEPTH).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// STRIDE = 0: a tile is TILE contiguous bytes.  STRIDE > 0: a tile is TILE/128 rows of 128 B at a row pitch of STRIDE bytes
// (a k-tile of a row-major [N][K] matrix with K*2 = STRIDE), consecutive tiles advance by 128 B along the row.
template <int TILE, int DEPTH, int STRIDE>
__global__ __launch_bounds__(512) void dma_kernel(const char* __restrict__ w, long wbytes, int ntiles, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int PER = TILE / 1024 / 8;   // DMA instructions per wave per tile
    float accv = 0.f;
    auto issue = [&](int t) {
        char* slot = smem + (t % DEPTH) * TILE;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int blk = wave * PER + i;
            long src;
            if (STRIDE == 0) src = ((long)t * TILE) % wbytes + blk * 1024 + lane * 16;
            else {
                const int row = blk * 8 + (lane >> 3);
                const int ktiles = STRIDE / 128;                       // k-tiles per row panel
                const long panel = (long)(t / ktiles) * (TILE / 128) * STRIDE;   // next group of rows after a full K sweep
                src = (panel % wbytes) + (long)row * STRIDE + (t % ktiles) * 128 + (lane & 7) * 16;
            }
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(w + src), (lds_void_t*)(slot + blk * 1024), 16, 0, 0);
        }
    };
    for (int t = 0; t < DEPTH - 1 && t < ntiles; ++t) issue(t);
    for (int t = 0; t < ntiles; ++t) {
        // oldest tile t must have landed: at most (DEPTH-2) younger tiles (PER instr each) may remain in flight
        if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (DEPTH == 3) { if (PER == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else if (PER == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else if (PER == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        else if (DEPTH == 4) { if (PER == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else if (PER == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else if (PER == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
        else { if (PER == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else if (PER == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        if (t + DEPTH - 1 < ntiles) issue(t + DEPTH - 1);   // refills the slot consumed in iteration t-1
        accv += *reinterpret_cast<const float*>(smem + (t % DEPTH) * TILE + ((tid * 16) % TILE));
    }
    if (accv == 123.456f) out[0] = accv;
}

template <int TILE, int DEPTH, int STRIDE = 0>
void run(const char* w, long wbytes, float* out, int ncu) {
    const int ntiles = (int)(8L * 1024 * 1024 / TILE);   // 8 MB per workgroup
    hipFuncSetAttribute((const void*)dma_kernel<TILE, DEPTH, STRIDE>, hipFuncAttributeMaxDynamicSharedMemorySize, TILE * DEPTH);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL((dma_kernel<TILE, DEPTH, STRIDE>), dim3(ncu), dim3(512), TILE * DEPTH, 0, w, wbytes, ntiles, out);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("stride %4d  tile %3d KB depth %d (in flight %3d KB, LDS %3d KB): %6.1f GB/s per CU, %6.2f us per tile\n", STRIDE, TILE / 1024, DEPTH,
           TILE * (DEPTH - 1) / 1024, TILE * DEPTH / 1024, 8.0 * 1024 * 1024 / (ms * 1e-3) / 1e9, ms * 1e3 / ntiles);
}

int main() {
    const long wbytes = 3L << 20;
    char* w; float* out;
    hipMalloc(&w, wbytes); hipMemset(w, 1, wbytes); hipMalloc(&out, 64);
    const int ncu = 225;
    run<32768, 2, 1024>(w, wbytes, out, ncu); run<32768, 2, 2048>(w, wbytes, out, ncu); run<16384, 2, 1024>(w, wbytes, out, ncu);
    run<65536, 2, 1024>(w, wbytes, out, ncu); run<32768, 3, 1024>(w, wbytes, out, ncu);
    run<32768, 2>(w, wbytes, out, ncu); run<16384, 2>(w, wbytes, out, ncu); run<16384, 3>(w, wbytes, out, ncu); run<16384, 4>(w, wbytes, out, ncu);
    run<8192, 4>(w, wbytes, out, ncu); run<8192, 8>(w, wbytes, out, ncu); run<32768, 3>(w, wbytes, out, ncu); run<32768, 4>(w, wbytes, out, ncu);
    run<65536, 2>(w, wbytes, out, ncu); run<16384, 8>(w, wbytes, out, ncu);
    return 0;
}
