// K/V-resident attention kernel of the TCDiff denoiser (bf16, gfx950), shared by attention.hip (inference) and
// attention_train.hip (train-mode forward: + dropout + lse).
#pragma once
#include "attn_common.h"
#include "train_common.h"

// =================================================================================================
// K/V-resident variant (bf16): workgroups of 8 waves per (sequence, head, block of 256 or 512 query rows).
// The streaming kernel above is latency-bound per 64-key tile (stamped: ~6500 cycles per tile for 512 cycles of MFMA:
// LDS-read -> MFMA chains, one barrier per tile, 32 query rows of independent work per wave).  For the denoiser's
// shapes (L = 450 tokens, 152 memory rows) all of K and V of one (sequence, head) fits in LDS (2 x 64 KB), so this
// kernel loads them ONCE (LDS-DMA, one barrier), gives every wave 64 query rows as two independent 32-row groups that
// share every K / V^T fragment read, and runs the whole key loop without barriers.  At B = 16 that is 2*16*8 = 256
// workgroups: one per CU.
// =================================================================================================
#define ATT_RES_MAXT 8   // up to 8 tiles of 64 keys

// TRAIN (attention_train.hip): dropout on the softmax weights (counter hash of train_common.h, flat index of the element in
// the reference's [n_seq * H, Lq, Lk] weights tensor) and the row statistic lse = m + log2(l) for the backward pass.
struct AttnTrainArgs {
    float* lse;            // [n_seq * H][Lp_q]
    const int* seed;
    int site;
    uint32_t thr;
    float dscale;
    char* o_lo;            // optional second output image: bf16(o - float(bf16(o))) (tcdiff_attention_train)
};

template <int NG, bool TRAIN>
__global__ __launch_bounds__(512) void attention_res_kernel(const char* __restrict__ Q, const char* __restrict__ K,
                                                            const char* __restrict__ V, char* __restrict__ O, int H, int Lq,
                                                            int Lk, int Lp_q, int Lp_k, int ldo, int n_shared,
                                                            AttnTrainArgs ta) {
    typedef MmaBF16 P;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    // NG row groups of 32 per wave: a workgroup covers 256 * NG query rows; with NG = 1 a 450-token sequence is two
    // workgroups (each loads all of K and V), which puts a half-batch launch on every CU instead of half of them
    const int nqb = (Lq + 256 * NG - 1) / (256 * NG);
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = wg % nqb, head = (wg / nqb) % H, seq = wg / (nqb * H);
    const int kv = seq < n_shared ? 0 : seq - n_shared + (n_shared > 0 ? 1 : 0);
    const int ntm = (Lk + 63) / 64 < ATT_RES_MAXT ? (Lk + 63) / 64 : ATT_RES_MAXT;   // tiles of the largest key chunk
    char* Ks = smem;                       // [ntm][64 keys][128 B], chunk-swizzled (common.h tile_off)
    char* Vs = smem + ntm * 8192;
    const char* Kg = K + (long)(kv * H + head) * Lp_k * 128;
    const char* Vg = V + (long)(kv * H + head) * Lp_k * 128;
    // ---- Q^T fragments of the NG row groups (registers for the whole kernel)
    const int qbase = qb * 256 * NG + wave * 32 * NG;
    u32x4 qf[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        int qrow = qbase + g * 32 + r;
        qrow = qrow < Lp_q ? qrow : Lp_q - 1;          // rows past the padded image belong to inactive groups
        const char* Qg = Q + ((long)(seq * H + head) * Lp_q + qrow) * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[g][ks] = *reinterpret_cast<const u32x4*>(Qg + (2 * ks + h) * 16);
    }
    const bool act0 = qbase < Lq, act1 = NG > 1 && qbase + 32 < Lq;   // wave-uniform
    f32x16_t o[NG][2];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int q = 0; q < 16; ++q) o[g][dt][q] = 0.0f;
    float m_run[NG], l_run[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) { m_run[g] = -INFINITY; l_run[g] = 0.0f; }
    DropCtx dc = {};
    if constexpr (TRAIN) dc = drop_ctx(ta.seed, ta.site, ta.thr, ta.dscale);

    // keys in LDS-resident chunks of up to 512 (ONE chunk when L <= 512): a chunk is loaded by LDS-DMA, then its tiles
    // run barrier-free; longer sequences (config 4: L = 1500) pay one barrier pair per chunk
    for (int c0 = 0; c0 < Lk; c0 += 64 * ATT_RES_MAXT) {
    const int nt = (Lk - c0 + 63) / 64 < ATT_RES_MAXT ? (Lk - c0 + 63) / 64 : ATT_RES_MAXT;
    if (c0 > 0) __syncthreads();           // every wave is done with the previous chunk
#ifndef ATT_ABLATE_LOAD   // (timing experiment: the launch without its K / V loads -- results wrong)
    for (int blk = wave; blk < nt * 8; blk += 8) {      // blk = 8 consecutive keys of the chunk
        const int row = blk * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ tile_swz(row);
        typedef __attribute__((address_space(3))) void lds_void_t;
        typedef const __attribute__((address_space(1))) void gbl_void_t;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(Kg + (long)(c0 + row) * 128 + chunk * 16), (lds_void_t*)(Ks + blk * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(Vg + (long)(c0 + row) * 128 + chunk * 16), (lds_void_t*)(Vs + blk * 1024), 16, 0, 0);
    }
#endif
    // (Tried and dropped: replacing this full drain by counted waits, `s_waitcnt vmcnt(2 (nt - 1 - b))` + barrier in front of
    // tile b, so that tile 0 starts one tile's worth of DMA after the launch.  The backend's wait-count pass treats an LDS-DMA
    // in flight as aliasing EVERY later ds_read and puts its own `s_waitcnt vmcnt(0)` in front of the first K fragment read
    // of each tile -- visible in the ISA -- so the drain happens anyway, one tile later.  Avoiding it needs every LDS read of
    // the tile in inline asm with hand-counted lgkmcnt; not worth it for ~4 us of a 28-us launch.
    // Round 4 built the other way round it: EVERY vector-memory operation of the launch (16 LDS-DMA + the 8 Q loads) issued
    // from one asm statement the compiler cannot see into, `s_waitcnt vmcnt(12)` + barrier in front of tiles 0-1, a second
    // wait + barrier in front of the other six; ISA as intended (no compiler-inserted drain), results identical -- and the same
    // time: 32.0 against 32.4 us in the microbenchmark, 14.0 / 14.1 against 14.1 / 14.6 clips/s in the sampler (same box,
    // interleaved).  The prologue is ISSUE-bound, not latency-bound: a wave needs ~200 cycles per DMA instruction, 24 of them
    // are ~2.4 us during which it computes nothing whatever the wait that follows; hiding that means issuing tile b + 2 inside
    // tile b's work, i.e. a wait + barrier per tile, which is the streaming kernel this one replaced.)
    sync_dma();

    if (act0) {
        // one tile of NS * 32 keys (NS = 2 except for a last tile with <= 32 keys left: 450 keys = 7 tiles + 2 keys)
        auto tile = [&](auto ns_c, int b) {
            constexpr int NS = decltype(ns_c)::value;
            const char* kt_base = Ks + b * 8192;
            const char* vt_base = Vs + b * 8192;
            const int kv0 = c0 + b * 64;
            // ---- S^T = K Q^T for both row groups: every K fragment read feeds two MFMAs
            f32x16_t s[NG][2];
#pragma unroll
            for (int kt = 0; kt < NS; ++kt) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const u32x4 kf = *reinterpret_cast<const u32x4*>(kt_base + tile_off(kt * 32 + r, 2 * ks + h));
                    if (ks == 0) {   // C = literal 0: the MFMA takes the inline constant, no 64 v_mov per tile
                        const f32x16_t z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                        for (int g = 0; g < NG; ++g) s[g][kt] = z;
                    }
#pragma unroll
                    for (int g = 0; g < NG; ++g) P::mma(s[g][kt], kf, qf[g][ks]);
                }
            }
            if (kv0 + 64 > Lk) {
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int kt = 0; kt < NS; ++kt)
#pragma unroll
                        for (int q = 0; q < 16; ++q)
                            if (kv0 + kt * 32 + acc_row(q, h) >= Lk) s[g][kt][q] = -INFINITY;
            }
            constexpr float LOG2E = 1.4426950408889634f;
            if constexpr (!TRAIN) {
            // ---- online softmax with the exponentials INSIDE the PV MFMA stream.  On a SIMD the VALU work of one wave does not run
            // under the MFMAs of the other (tools/probe/coissue_probe.hip); what hides is single-issue VALU
            // work in a wave's OWN stream, ~24 issue cycles per 32x32x16 MFMA, and v_exp_f32 (8 cycles, no packed form) is the ideal
            // filler.  So: row maximum, rescale, x = s log2e - m in packed math as before -- then the PV phase walks its units (key
            // tile, 16-key step, row group: eight per 64-key tile with two row groups), each feeding two MFMAs, and the eight exponentials + four packs of unit u + 1 are
            // issued behind the MFMAs of unit u (three exponentials per MFMA; the last two and the packs in the open).  The same
            // instructions as before, moved (bit-identical results); the row sums follow the PV phase.  Measured, 256 workgroups,
            // same box, three interleaved repetitions: 32.2 -> 30.9 us per launch (1 / 2 / 3 / 4 exponentials per MFMA: 31.4 / 31.1 /
            // 30.9 / 31.0).  Also built: one exponential + four plain v_add_f32 of the row sums per MFMA (31.7: the unpacked sums are
            // twice the instructions of the packed ones they replace); group 0's eight S MFMAs first and its row maximum + scaling,
            // unpacked, behind group 1's (K fragments read twice: 2 % slower).
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                float mx = s[g][0][0];
#pragma unroll
                for (int kt = 0; kt < NS; ++kt)
#pragma unroll
                    for (int q = 0; q < 16; ++q) mx = fmaxf(mx, s[g][kt][q]);
                mx = fmaxf(mx, other_half(mx)) * LOG2E;
                const float m_new = fmaxf(m_run[g], mx);
                const f32x2_t l2 = {LOG2E, LOG2E}, nm = {-m_new, -m_new};
#pragma unroll
                for (int kt = 0; kt < NS; ++kt)
#pragma unroll
                    for (int q = 0; q < 16; q += 2) {
                        const f32x2_t x = __builtin_elementwise_fma(f32x2_t{s[g][kt][q], s[g][kt][q + 1]}, l2, nm);
                        s[g][kt][q] = x[0];
                        s[g][kt][q + 1] = x[1];
                    }
                if (__builtin_amdgcn_ballot_w64(m_new > m_run[g]) != 0) {
                    const float alpha = __builtin_amdgcn_exp2f(m_run[g] - m_new);
                    l_run[g] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int q = 0; q < 16; ++q) o[g][dt][q] *= alpha;
                    m_run[g] = m_new;
                }
            }
            constexpr int NU = 2 * NS * NG;             // units u = (key tile, 16-key step, row group): g fastest
            u32x4 pf[2];
            auto ex = [&](int u, int j) {               // exponential j of unit u, in place
                const int g = u % NG, kt = u / (2 * NG), q = 8 * ((u / NG) & 1) + j;
                s[g][kt][q] = __builtin_amdgcn_exp2f(s[g][kt][q]);
            };
            auto packs = [&](int u) {
                const int g = u % NG, kt = u / (2 * NG), q0 = 8 * ((u / NG) & 1);
#pragma unroll
                for (int i = 0; i < 4; ++i) pf[u & 1][i] = pack_bf2(s[g][kt][q0 + 2 * i], s[g][kt][q0 + 2 * i + 1]);
            };
            constexpr int EPG = 3;                      // exponentials of unit u + 1 behind each MFMA of unit u
#pragma unroll
            for (int j = 0; j < 8; ++j) ex(0, j);
            packs(0);
            u32x4 vf[2];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int g = u % NG, kt = u / (2 * NG), st = (u / NG) & 1;
                if (g == 0) {
                    vf[0] = v_frag<P>(vt_base, 0, kt, st, lane);
                    vf[1] = v_frag<P>(vt_base, 1, kt, st, lane);
                }
                __builtin_amdgcn_sched_barrier(0);
                P::mma(o[g][0], vf[0], pf[u & 1]);
                if (u + 1 < NU) {
#pragma unroll
                    for (int j = 0; j < EPG; ++j) ex(u + 1, j);
                }
                __builtin_amdgcn_sched_barrier(0);
                P::mma(o[g][1], vf[1], pf[u & 1]);
                if (u + 1 < NU) {
#pragma unroll
                    for (int j = EPG; j < 2 * EPG; ++j) ex(u + 1, j);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (u + 1 < NU) {
#pragma unroll
                    for (int j = 2 * EPG; j < 8; ++j) ex(u + 1, j);
                    packs(u + 1);
                }
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                f32x2_t rs2 = {0.0f, 0.0f};
#pragma unroll
                for (int kt = 0; kt < NS; ++kt)
#pragma unroll
                    for (int q = 0; q < 16; q += 2) rs2 += f32x2_t{s[g][kt][q], s[g][kt][q + 1]};
                float rs = rs2[0] + rs2[1];
                rs += other_half(rs);
                l_run[g] += rs;
            }
            } else {
            // ---- online softmax (base 2), the two groups are independent instruction streams
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                float mx = s[g][0][0];
#pragma unroll
                for (int kt = 0; kt < NS; ++kt)
#pragma unroll
                    for (int q = 0; q < 16; ++q) mx = fmaxf(mx, s[g][kt][q]);
                mx = fmaxf(mx, other_half(mx)) * LOG2E;
                const float m_new = fmaxf(m_run[g], mx);
                // x = s * log2(e) - m and the row sum as float2 ops (v_pk_fma_f32 / v_pk_add_f32); exp2 stays scalar
                f32x2_t rs2 = {0.0f, 0.0f};
                const f32x2_t l2 = {LOG2E, LOG2E}, nm = {-m_new, -m_new};
#pragma unroll
                for (int kt = 0; kt < NS; ++kt)
#pragma unroll
                    for (int q = 0; q < 16; q += 2) {
                        const f32x2_t x = __builtin_elementwise_fma(f32x2_t{s[g][kt][q], s[g][kt][q + 1]}, l2, nm);
#ifdef ATT_ABLATE_EXP   // (timing experiment: no transcendental)
                        const f32x2_t p = x;
#else
                        const f32x2_t p = {__builtin_amdgcn_exp2f(x[0]), __builtin_amdgcn_exp2f(x[1])};
#endif
                        s[g][kt][q] = p[0];
                        s[g][kt][q + 1] = p[1];
                        rs2 += p;                               // the softmax denominator is taken BEFORE the dropout
                    }
                if constexpr (TRAIN) {
                    // dropout as its own pass over the registers, under ONE wave-uniform branch: inside the exp loop it broke that
                    // loop's packed-math schedule (p = 0: 37 us against the sampler's 28 us for the same work)
                    if (ta.thr) {
                        const uint32_t x0 = ((uint32_t)(seq * H + head) * (uint32_t)Lq + (uint32_t)(qbase + g * 32 + r)) * (uint32_t)Lk +
                                            (uint32_t)kv0;
#pragma unroll
                        for (int kt = 0; kt < NS; ++kt)
#pragma unroll
                            for (int q = 0; q < 16; ++q)
                                s[g][kt][q] = drop_apply(dc, x0 + (uint32_t)(kt * 32 + acc_row(q, h)), s[g][kt][q]);
                    }
                }
                float rs = rs2[0] + rs2[1];
                rs += other_half(rs);
                if (__builtin_amdgcn_ballot_w64(m_new > m_run[g]) != 0) {
                    const float alpha = __builtin_amdgcn_exp2f(m_run[g] - m_new);
                    l_run[g] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                        for (int q = 0; q < 16; ++q) o[g][dt][q] *= alpha;
                    m_run[g] = m_new;
                }
                l_run[g] += rs;
            }
            // ---- O^T += V^T P^T: every V^T fragment (two transposed reads) feeds both groups
#pragma unroll
            for (int kt = 0; kt < NS; ++kt)
#pragma unroll
                for (int st = 0; st < 2; ++st) {
                    u32x4 pf[NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        pf[g][0] = pack_bf2(s[g][kt][8 * st + 0], s[g][kt][8 * st + 1]);
                        pf[g][1] = pack_bf2(s[g][kt][8 * st + 2], s[g][kt][8 * st + 3]);
                        pf[g][2] = pack_bf2(s[g][kt][8 * st + 4], s[g][kt][8 * st + 5]);
                        pf[g][3] = pack_bf2(s[g][kt][8 * st + 6], s[g][kt][8 * st + 7]);
                    }
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const u32x4 vf = v_frag<P>(vt_base, dt, kt, st, lane);
#pragma unroll
                        for (int g = 0; g < NG; ++g) P::mma(o[g][dt], vf, pf[g]);
                    }
                }
            }
        };
        // (Measured and dropped, round 3: the same tile as a software pipeline over the wave's two row groups -- S(g1) under
        // the exp stream of g0, PV(g0) under the exp stream of g1, placed with sched_group_barrier; the ISA interleaves as
        // asked, 246 VGPRs, no scratch -- 29.5 us per launch against 28.4: the SIMD's second wave already fills those gaps.)
        const int nfull = (Lk - c0 - 64 * (nt - 1)) <= 32 ? nt - 1 : nt;     // tiles that use both 32-key halves
        // (Also measured and dropped: delaying waves 4-7 by ~half a tile (s_sleep 10 / 19 / 28) so that the SIMD's two waves are
        // in different phases -- MI355X_MICROARCH.md, two waves per SIMD, item 9: within the +-1 % noise of the same box.)
#pragma unroll 1
        for (int b = 0; b < nfull; ++b) tile(std::integral_constant<int, 2>{}, b);
        if (nfull < nt) tile(std::integral_constant<int, 1>{}, nt - 1);
    }
    }  // key chunks
    // ---- O[q][d] = O^T[d][q] / l.  A lane holds 8 bytes of a row at a time; stored like that an instruction makes 32
    // sixteen-byte write requests.  Each 32-row group goes through 4 KB of wave-private LDS behind the K / V images
    // (XOR-swizzled by (row >> 1) & 7: rows alternate between the two 128-byte halves of the 64 banks; a wave's LDS queue is in order, no barrier) and leaves as 16 bytes per lane, 8 lanes per
    // 128-byte row: 8 full lines per store instruction.
    char* stg = smem + 2 * ntm * 8192 + wave * 4096;
    const int srow0 = lane >> 3, sch = lane & 7;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (!(g == 0 ? act0 : act1)) continue;      // wave-uniform
        const float inv = 1.0f / l_run[g];
        if constexpr (TRAIN) {
            const int qg = qbase + g * 32 + r;
            if (h == 0 && qg < Lq) ta.lse[(long)(seq * H + head) * Lp_q + qg] = m_run[g] + log2f(l_run[g]);
        }
        // (train mode: a second pass stores what the 8-bit image dropped, bf16(o - float(bf16(o))): AttnTrainArgs.o_lo)
        bool lo_pass = false;
        if constexpr (TRAIN) lo_pass = ta.o_lo != nullptr;
        for (int pass = 0; pass < (lo_pass ? 2 : 1); ++pass) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = o[g][dt][4 * q4 + j] * inv;
                    if (pass == 1) v[j] -= __builtin_bit_cast(float, pack_bf2(v[j], 0.0f) << 16);
                }
                uint2 pk;
                pk.x = pack_bf2(v[0], v[1]);
                pk.y = pack_bf2(v[2], v[3]);
                *reinterpret_cast<uint2*>(stg + r * 128 + (((4 * dt + q4) ^ ((r >> 1) & 7)) << 4) + 8 * h) = pk;
            }
        char* dstO = O;
        if constexpr (TRAIN) dstO = pass == 1 ? ta.o_lo : O;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = srow0 + 8 * k;
            const u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * 128 + ((sch ^ ((row >> 1) & 7)) << 4));
            const int qg = qbase + g * 32 + row;
#ifdef ATT_ABLATE_OSTORE
            if (v.x == 0x12345678u)
#endif
            if (qg < Lq)
                *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(dstO) + ((long)seq * Lq + qg) * ldo + head * 64 + sch * 8) = v;
        }
        }
    }
}

