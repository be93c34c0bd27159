// Probe (round 6): what does a release / flag / acquire exchange among the 4 workgroups of one 16-row block cost inside ONE launch,
// against the launch boundary the small-job layer uses today?  116 workgroups x 512 threads, 160 KB of LDS each (one per CU),
// ROUNDS exchanges: every member stores its 32 KB partial slab, releases, bumps the group's counter, spins (bounded) until all four
// have, acquires, reads the four slabs.  Prints us per round for: flags only / flags + data, and for the same data exchange as
// back-to-back launches.
//   hipcc --offload-arch=gfx950 -O3 -o flag_sync flag_sync.hip && ./flag_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, rem = nblocks & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
}

template <bool DATA, bool REMAP>
__global__ __launch_bounds__(512) void sync_kernel(float* P, unsigned* flags, int rounds, unsigned long long* t_out, unsigned* err) {
    extern __shared__ char smem[];
    const int logical = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int grp = logical >> 2, member = logical & 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, gg = lane >> 4;
    f32x4 acc[4];
    for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{1.0f * member, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < rounds; ++r) {
        float* slab = P + (long)(r & 1) * gridDim.x * 16 * 512;
        if (DATA) {
            float* dst = slab + ((long)(grp * 4 + member) * 16 + c) * 512 + 64 * wave + 4 * gg;
            for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4*>(dst + 16 * nt) = acc[nt];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(flags + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = 4u * (r + 1);
            int spins = 0;
            while (__hip_atomic_load(flags + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++spins > (1 << 22)) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (DATA) {
            const float* src = slab + ((long)(grp * 4) * 16 + c) * 512 + 64 * wave + 4 * gg;
            f32x4 s[4] = {};
            for (int m = 0; m < 4; ++m)
                for (int nt = 0; nt < 4; ++nt) s[nt] += *reinterpret_cast<const f32x4*>(src + (long)m * 16 * 512 + 16 * nt);
            for (int nt = 0; nt < 4; ++nt) acc[nt] = s[nt] * 0.25f + f32x4{1.0f * member, 0, 0, 0};
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) t_out[blockIdx.x] = t1 - t0;
    if (DATA && lane == 0 && wave == 0) P[2L * gridDim.x * 16 * 512 + blockIdx.x] = acc[0][0];     // keep the sums alive; expect 1.5 * ... a fixed point
}

// The same exchange WITHOUT fences: the slabs are written and read with scope bits on the accesses themselves (SC = 1: sc1, agent;
// SC = 2: sc0 sc1, system; SC = 3: sc0 only = workgroup scope / L1 bypass: valid only if the four members share an L2), the flag
// with relaxed agent-scope atomics; s_waitcnt vmcnt(0) orders a member's stores before its flag bump.  Every round's values differ
// (member + round), and every member checks the sum it reads: stale data is counted.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// raw-buffer accesses with cache-policy bits the compiler can see (it places the s_waitcnt): aux bit 0 = sc0, bit 4 = sc1 on gfx94x/95x
template <int SC> __device__ inline void st4(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, off, 0, SC == 1 ? 16 : SC == 2 ? 17 : 1);
}
template <int SC> __device__ inline f32x4 ld4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, SC == 1 ? 16 : SC == 2 ? 17 : 1));
}
template <int SC, bool REMAP>
__global__ __launch_bounds__(512) void sync_nofence_kernel(float* P, unsigned* flags, int rounds, unsigned long long* t_out, unsigned* err) {
    extern __shared__ char smem[];
    const int logical = REMAP ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int grp = logical >> 2, member = logical & 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, gg = lane >> 4;
    unsigned bad = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < rounds; ++r) {
        float* slab = P + (long)(r & 1) * gridDim.x * 16 * 512;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slab, 0, (int)(gridDim.x * 16 * 512 * 4), 0x00020000);
        const unsigned dst = (((grp * 4 + member) * 16 + c) * 512 + 64 * wave + 4 * gg) * 4;
        const float val = (float)(member + r);
        for (int nt = 0; nt < 4; ++nt) st4<SC>(rs, dst + 64 * nt, f32x4{val, val, val, val});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(flags + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = 4u * (r + 1);
            int spins = 0;
            while (__hip_atomic_load(flags + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++spins > (1 << 22)) { *err = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        const unsigned src = (((grp * 4) * 16 + c) * 512 + 64 * wave + 4 * gg) * 4;
        f32x4 v[16];
        for (int m = 0; m < 4; ++m)
            for (int nt = 0; nt < 4; ++nt) v[m * 4 + nt] = ld4<SC>(rs, src + m * 16 * 512 * 4 + 64 * nt);
        for (int nt = 0; nt < 4; ++nt) {
            const f32x4 s = v[nt] + v[4 + nt] + v[8 + nt] + v[12 + nt];
            for (int k = 0; k < 4; ++k) bad += s[k] != (float)(4 * r + 6);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) t_out[blockIdx.x] = t1 - t0;
    if (bad) atomicAdd(err + 1, bad);
}
template <int SC, bool REMAP>
static int run_nf(const char* name, int grid, int rounds, float* P, unsigned* flags, unsigned long long* t, unsigned* err) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(sync_nofence_kernel<SC, REMAP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    float best = 1e9f;
    std::vector<unsigned long long> ht(grid);
    unsigned long long worst = 0;
    CK(hipMemset(err, 0, 8));
    for (int it = 0; it < 5; ++it) {
        CK(hipMemset(flags, 0, 4096));
        CK(hipEventRecord(e0));
        sync_nofence_kernel<SC, REMAP><<<grid, 512, 160 * 1024>>>(P, flags, rounds, t, err);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) {
            best = ms;
            CK(hipMemcpy(ht.data(), t, grid * 8, hipMemcpyDeviceToHost));
            worst = 0;
            for (auto v : ht) worst = v > worst ? v : worst;
        }
    }
    unsigned herr[2]; CK(hipMemcpy(herr, err, 8, hipMemcpyDeviceToHost));
    printf("%-44s grid %3d: %7.3f us / round by events, %7.3f us / round inside; stale values read: %u%s\n", name, grid,
           best * 1e3f / rounds, worst / 100.0 / rounds, herr[1], herr[0] ? "  SPIN LIMIT HIT" : "");
    return 0;
}

__global__ __launch_bounds__(512) void launch_kernel(const float* Pin, float* Pout) {
    extern __shared__ char smem[];
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int grp = logical >> 2, member = logical & 3;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, gg = lane >> 4;
    const float* src = Pin + ((long)(grp * 4) * 16 + c) * 512 + 64 * wave + 4 * gg;
    f32x4 s[4] = {};
    for (int m = 0; m < 4; ++m)
        for (int nt = 0; nt < 4; ++nt) s[nt] += *reinterpret_cast<const f32x4*>(src + (long)m * 16 * 512 + 16 * nt);
    float* dst = Pout + ((long)(grp * 4 + member) * 16 + c) * 512 + 64 * wave + 4 * gg;
    for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4*>(dst + 16 * nt) = s[nt] * 0.25f + f32x4{1.0f * member, 0, 0, 0};
}

template <bool DATA, bool REMAP>
static int run(const char* name, int grid, int rounds, float* P, unsigned* flags, unsigned long long* t, unsigned* err) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(sync_kernel<DATA, REMAP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    float best = 1e9f;
    std::vector<unsigned long long> ht(grid);
    unsigned long long worst = 0;
    for (int it = 0; it < 5; ++it) {
        CK(hipMemset(flags, 0, 4096));
        CK(hipEventRecord(e0));
        sync_kernel<DATA, REMAP><<<grid, 512, 160 * 1024>>>(P, flags, rounds, t, err);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) {
            best = ms;
            CK(hipMemcpy(ht.data(), t, grid * 8, hipMemcpyDeviceToHost));
            worst = 0;
            for (auto v : ht) worst = v > worst ? v : worst;
        }
    }
    unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("%-44s grid %3d: %7.3f us / round by events, %7.3f us / round inside the kernel (slowest workgroup)%s\n", name, grid,
           best * 1e3f / rounds, worst / 100.0 / rounds, herr ? "  SPIN LIMIT HIT" : "");
    return 0;
}

int main() {
    const int grid = 116, rounds = 200;
    float* P; unsigned* flags; unsigned long long* t; unsigned* err;
    CK(hipMalloc(&P, (2L * 256 * 16 * 512 + 1024) * 4));
    CK(hipMemset(P, 0, (2L * 256 * 16 * 512 + 1024) * 4));
    CK(hipMalloc(&flags, 4096)); CK(hipMalloc(&t, 256 * 8)); CK(hipMalloc(&err, 8)); CK(hipMemset(err, 0, 8));
    if (run<false, true>("flags only, members share an XCD", grid, rounds, P, flags, t, err)) return 1;
    if (run<true, true>("flags + 32 KB out / 128 KB in, shared XCD", grid, rounds, P, flags, t, err)) return 1;
    if (run<false, false>("flags only, members on 4 different XCDs", grid, rounds, P, flags, t, err)) return 1;
    if (run<true, false>("flags + data, members on 4 different XCDs", grid, rounds, P, flags, t, err)) return 1;
    if (run<true, true>("flags + data, shared XCD, 232 workgroups", 232, rounds, P, flags, t, err)) return 1;
    if (run_nf<1, true>("no fences, sc1 accesses, shared XCD", grid, rounds, P, flags, t, err)) return 1;
    if (run_nf<1, false>("no fences, sc1 accesses, 4 XCDs", grid, rounds, P, flags, t, err)) return 1;
    if (run_nf<2, true>("no fences, sc0 sc1 accesses, shared XCD", grid, rounds, P, flags, t, err)) return 1;
    if (run_nf<2, false>("no fences, sc0 sc1 accesses, 4 XCDs", grid, rounds, P, flags, t, err)) return 1;
    if (run_nf<3, true>("no fences, sc0 accesses, shared XCD", grid, rounds, P, flags, t, err)) return 1;
    if (run_nf<3, false>("no fences, sc0 accesses, 4 XCDs (expect stale)", grid, rounds, P, flags, t, err)) return 1;
    if (run_nf<1, true>("no fences, sc1, shared XCD, 232 workgroups", 232, rounds, P, flags, t, err)) return 1;
    // the same exchange as launches (graph of 200 kernel nodes)
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(launch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipGraph_t gr; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int r = 0; r < rounds; ++r) {
        float* a = P + (long)(r & 1) * 256 * 16 * 512; float* b = P + (long)((r + 1) & 1) * 256 * 16 * 512;
        launch_kernel<<<grid, 512, 160 * 1024, s>>>(a, b);
    }
    CK(hipStreamEndCapture(s, &gr));
    CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
    }
    printf("%-44s grid %3d: %7.3f us / launch (captured graph of %d launches)\n", "the same exchange at launch boundaries", grid, best * 1e3f / rounds, rounds);
    return 0;
}
