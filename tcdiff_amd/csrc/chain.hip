// Row-block chains of the TCDiff decoder layer (bf16 throughput mode), gfx950.
//
// Everything between the two attentions of a FiLMTransformerDecoderLayer is ROW-LOCAL (a token row needs only itself,
// the weights and its sequence's FiLM vectors): model/model.py:103-106,327 (fc + LayerNorm + FiLM + residual),
// :332,387 (norm2 + rotary + w_qs), and :334,338-339,344,399-401 followed by the next layer's :326,374-383
// (fc + LN + FiLM + residual, norm3, linear1 + GELU, linear2 + FiLM + residual, norm4, linear3, norm1 + rotary,
// w_qs/w_ks/w_vs).  The op-by-op kernels (gemm.hip) move every intermediate through HBM/MALL: ~38 KB per row and layer
// and nine launches.  Here one workgroup keeps a block of 64 rows on the CU for a whole chain:
//
//   chain A : O_self --fc--> LN(1e-6), FiLM, +x --> x (fp32, HBM)  --norm2, rotary--> [LDS] --w_qs--> Q image
//   chain B : O_cross --fc--> LN, FiLM, +x --> x (HBM) --norm3--> [LDS] --linear1, GELU--> [LDS, 256-column chunks]
//             --linear2--> FiLM, +x --norm4--> [LDS] --linear3--> x' (fp32, HBM) --norm1', rotary--> [LDS]
//             --w_qs / w_ks--> Q, K images ; norm1' --> [LDS] --w_vs--> V image        (last layer: stops after linear3)
//   FULL    : chain A, then the cross-attention itself (head w on wave w; K / V from fragment-ordered cache images, the
//             Q^T accumulator tiles are the B operand of the score MFMA), then chain B: ONE launch per decoder layer.
//             This is the production path; A and B alone are the reference it is tested against.
//
// Only the weights stream.  Structure:
//   * 8 waves, wave w owns output columns [64 w, 64 w + 64) of every 512-wide GEMM (= head w of Q / K / V) for all 64
//     rows.  Round 4: the products are v_mfma_f32_16x16x32_bf16 (round 1-3: 32x32x16).  Same cycles per FLOP, but every
//     accumulator register is read and written once per 32 deep k-step instead of once per 16, and the chip -- which this
//     launch holds at its power limit (225 busy CUs: 1.88-1.91 GHz against 2.39 GHz for a lone block, measured with
//     s_memtime) -- keeps a 7 % higher clock on it: the same launch took 100.4 instead of 111.0 us in the timing experiment
//     that led here (profiles/r04_chain_experiments.txt; cdna_hip_programming.md section 5.4 rule 28).
//   * MFMA operand roles are swapped (A operand = weight rows, B operand = activation rows): the accumulator tile
//     acc[nt][mt] holds, in lane l = 16 g + c, element j: column n = 64 w + 16 nt + 4 g + j of row m = 16 mt + c.  A lane
//     has 4 consecutive columns of 4 rows: LayerNorm statistics are in-register sums + a 3-swap reduce-scatter over the
//     four lane groups (v_permlane16_swap / v_permlane32_swap) + an 8-wave exchange through LDS, and FiLM / residual /
//     rotary / bf16 packing need no transposition.
//   * the weights of a chain are packed ON THE HOST (engine.py, once per checkpoint) into one linear stream per wave in
//     consumption order, in 4-KB stages that are already the MFMA fragment image ([n-tile 4][lane 64][16 B] of one 32-deep
//     k-step).  A wave consumes only fragments of its OWN columns, so weights never touch LDS: a stage is four coalesced
//     1-KB loads straight into registers, 4 stages (16 KB per wave, 128 KB per CU) are in flight in a register ring whose
//     slots are compile-time indices, and the GEMM phases (fully unrolled) have NO workgroup barrier and no
//     hand-written waits.  The stream runs ahead across GEMM and epilogue boundaries.
//   * activations live in LDS as [k-tile][64 rows][128 B] with the XOR chunk swizzle of common.h (tile_off).  Lane group g
//     of a B fragment takes the 16-byte chunk PI(g) = (0, 3, 1, 2)[g] of its 32-deep k-step (and the weight fragments are
//     packed with the same k order): with the natural order the two row sets a ds_read_b128 serves together ({0-3, 12-15}
//     of one chunk, {4-11} of the next) land on the same bank slots under the swizzle; chunks that differ by XOR 3 do not.
//   * every global access of an epilogue is made of contiguous 512-byte pieces: the fp32 residual stream and the
//     rotary table are column-blocked (RowPipe below), head-major images leave through wave-private LDS staging
//     (store_heads).
// Barriers: one per LayerNorm statistics exchange and one per activation hand-off.
#include "common.h"
#include "tcdiff_hip.h"

#include "chain_core.h"

#ifdef CH_ABLATE_ROWLAT      // timing experiment only (wrong values): the epilogues' residual / rotary rows 2, 3 are never fetched
#define CH_RP_NEXT(rp, nt) do { } while (0)
#else
#define CH_RP_NEXT(rp, nt) rp_issue(rp, nt)
#endif

// Cross-attention of this wave's head inside the chain (model/model.py:386-396,97-102 with cached K / V): the wave owns
// head `wave` of all 64 rows.  qacc = (rot W_q^T)^T tiles straight from the projection GEMM (lane = row, registers = d):
// scaled and packed, the pairs (nt = 2 s, 2 s + 1) ARE the B operand of S^T = K Q^T for the 32-deep d-step s, with d in the
// order the accumulators hold it (slot 8 g + jj <-> d = 32 s + 16 (jj >> 2) + 4 g + (jj & 3)) -- the K / V caches are kept
// in a second, fragment-ordered image (tcdiff_pack_kv_frags) whose 1-KB pieces load straight into the A operands with the
// same order.  Online softmax over 32-key tiles (two 16-key score tiles); O^T = V^T P^T with P^T fed from the S^T
// accumulator registers (key slot 8 g + jj <-> key 16 (jj >> 2) + 4 g + (jj & 3) of the tile).  The row sum stays a per-lane
// partial until the end; only the row maximum crosses the lane groups per tile.  Every K / V tile is loaded once and serves all
// four row tiles (the Q^T fragments wait in wave-private LDS meanwhile); a block whose rows straddle two sequences runs once
// per sequence and every lane keeps the result of its own row's sequence.
// Output: bf16 O rows into the activation block (columns 64 wave ..).
// HH: which of the wave's NT / 4 heads (NT = 4: head = wave; NT = 8: heads 2 wave, 2 wave + 1 = accumulator tiles 4 HH ..)
// SELF (round 5): the layer's SELF-attention with the same loop, in front of the fc GEMM (tcdiff_chain_args.sa_q): the Q^T
// fragments are the ones this block's previous launch packed (store_qfrag), the keys are the block's own sequence (K / V in
// fragment order from store_kfrag / store_vfrag of the previous launch, Lk = L, no slot mapping); blocks are cut per sequence.
// Mv: first row past the block's valid rows (M, or the end of the block's sequence); lblk: the block's logical index.
template <int HH, int MT, int NT, bool SELF = false>
DEVINL void cross_attention(const f32x4_t (&qacc)[NT][MT], const tcdiff_chain_args& a, int m0, int Mv, int lblk, char* abuf,
                            int wave, int lane) {
    // Cross-attention: passes of 32 rows, two row tiles (independent chains) per pass -- the weight ring is resident beside it.
    // SELF: all row tiles in ONE pass over the K / V tiles (the ring is loaded after it: the registers are there, and with 450 keys of
    // the block's own sequence a second pass doubles a K / V stream that already runs near the L2's rate when every CU does it at once).
    // CH_XATT_ONEPASS: diagnostic build (profiles/r05_chain_wave_forms.txt) -- one pass for the cross-attention too, Q^T parked in LDS.
#ifdef CH_XATT_ONEPASS
    constexpr bool PARK = !SELF;
#else
    constexpr bool PARK = false;
#endif
    constexpr bool ONEPASS = SELF || PARK;
    constexpr int NPS = ONEPASS ? 1 : (MT == 4 ? 2 : 1), NML = ONEPASS ? MT : (MT == 1 ? 1 : 2);
    lane = fresh_v(lane);
    wave = fresh_s(wave) * (NT / 4) + HH;          // from here on `wave` is the HEAD
    const int c = lane & 15, g = lane >> 4;
    const int M = Mv, L = a.L, nkt = SELF ? a.sa_nkt : a.nkt, Lk = SELF ? a.L : a.Lk;
    const float qs = a.scale_q * CH_LOG2E;         // the scores come out of the MFMAs in the exp2 domain
    u32x4 qf[MT][2];
    if constexpr (SELF) {
        const u32x4* qsrc = reinterpret_cast<const u32x4*>(a.sa_q) + ((long)(lblk * 8 + wave) * 8) * 64 + lane;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < 2; ++s) qf[mt][s] = qsrc[(mt * 2 + s) * 64];
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x4_t lo = qacc[4 * HH + 2 * s][mt], hi = qacc[4 * HH + 2 * s + 1][mt];
                qf[mt][s][0] = pack_bf2(lo[0] * qs, lo[1] * qs);
                qf[mt][s][1] = pack_bf2(lo[2] * qs, lo[3] * qs);
                qf[mt][s][2] = pack_bf2(hi[0] * qs, hi[1] * qs);
                qf[mt][s][3] = pack_bf2(hi[2] * qs, hi[3] * qs);
            }
    }
    // (PARK) the packed Q^T fragments wait in LDS (the GELU chunk area is idle here; 8 KB per wave, wave-private: no barrier)
    char* qpark = abuf + (CH_H1C - CH_ABUF) + wave * 8192 + lane * 16;
    if constexpr (PARK) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int s = 0; s < 2; ++s) *reinterpret_cast<u32x4*>(qpark + (mt * 2 + s) * 1024) = qf[mt][s];
    }
    u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};     // bf16 1.0 x 8: the A operand of the row-sum MFMA
    asm volatile("" : "+v"(ones));
    const unsigned voff = (unsigned)lane * 16u;
    const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(SELF ? a.sa_kf : a.kf), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(SELF ? a.sa_vf : a.vf), 0, -1, 0x00020000);
    auto ld_tile = [&](const __amdgpu_buffer_rsrc_t& r, unsigned so, u32x4 (&f)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff + 1024u * i, so, 0));
    };
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {             // rows 32 ps .. 32 ps + 31 = row tiles 2 ps, 2 ps + 1
        int ra = m0 + 16 * NML * ps, rb = ra + 16 * NML - 1;
        ra = ra < M ? ra : M - 1;
        rb = rb < M ? rb : M - 1;
        const int sa = ra / L, sb = rb / L;                 // wave-uniform
        int my_seq[NML];
#pragma unroll
        for (int ml = 0; ml < NML; ++ml) {
            int mrow = m0 + 16 * NML * ps + 16 * ml + c;
            mrow = mrow < M ? mrow : M - 1;
            my_seq[ml] = mrow / L;
        }
#pragma unroll 1
        for (int seq = sa; seq <= sb; ++seq) {
            const int kv = SELF ? seq : seq < a.n_shared ? 0 : seq - a.n_shared + (a.n_shared > 0 ? 1 : 0);
            const unsigned so0 = (unsigned)((kv * a.H + wave) * nkt) * 4096u;      // this (slot, head)'s image (< 4 GB: launcher)
            f32x4_t o[4][NML];                              // [d tile][row tile of the pass]
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int ml = 0; ml < NML; ++ml) o[dt][ml] = f32x4_t{0, 0, 0, 0};
            // Softmax bookkeeping that costs no VALU work in the common tile (the loop is bound by vector issue: 8 cycles per MFMA or
            // v_exp, 4 per other instruction, and they add):
            //  * the scores leave the MFMAs as s - m_run: the accumulators START at -m_run (nb), Q carries log2(e) / sqrt(d_k);
            //  * m_run is only an estimate of the row maximum: it moves when some score of the tile exceeds it by more than CH_ATT_THR
            //    (exp2 <= 2^THR: harmless in fp32 / bf16) -- a lane-local test, no cross-lane maximum in the common tile; the first
            //    tile always takes the exact path and sets m_run to its row maxima;
            //  * the row sums come from the matrix pipe: one more MFMA with an all-ones A operand sums the bf16 P^T columns (every row
            //    of its result is the sum over the tile's 32 keys: no cross-lane reduction at the end either).
            f32x4_t lacc[NML];
            float m_run[NML], nb[NML];
#pragma unroll
            for (int ml = 0; ml < NML; ++ml) {
                m_run[ml] = -INFINITY;
                nb[ml] = 0.0f;
                lacc[ml] = f32x4_t{0, 0, 0, 0};
            }
            u32x4 kn[4];                                           // K fragments run one tile ahead
            ld_tile(kr, so0, kn);
#pragma unroll 1
            for (int kt = 0; kt < nkt; ++kt) {
                u32x4 kc[4], vc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) kc[i] = kn[i];
                ld_tile(vr, so0 + (unsigned)kt * 4096u, vc);       // V of this tile: in flight under QK^T and the softmax
                const int nx = kt + 1 < nkt ? kt + 1 : kt;       // the last iteration re-reads its own tile (unused)
                if constexpr (!PARK) ld_tile(kr, so0 + (unsigned)nx * 4096u, kn);
                // the row tiles of the pass are independent chains in ONE basic block (no per-tile branch between them)
                f32x4_t s0[NML], s1[NML];                              // keys 4 g + j and 16 + 4 g + j of the tile
#pragma unroll
                for (int ml = 0; ml < NML; ++ml) {
                    const int mt = NML * ps + ml;
                    s0[ml] = s1[ml] = f32x4_t{nb[ml], nb[ml], nb[ml], nb[ml]};
                    u32x4 q0 = qf[mt][0], q1 = qf[mt][1];
                    if constexpr (PARK) {
                        char* qp = qpark;
                        asm volatile("" : "+v"(qp));      // opaque per tile: the reads stay in the loop (hoisted, they are 32 registers again)
                        q0 = *reinterpret_cast<const u32x4*>(qp + (mt * 2) * 1024);
                        q1 = *reinterpret_cast<const u32x4*>(qp + (mt * 2 + 1) * 1024);
                    }
                    mma16(s0[ml], kc[0], q0);
                    mma16(s1[ml], kc[2], q0);
                    mma16(s0[ml], kc[1], q1);
                    mma16(s1[ml], kc[3], q1);
                }
                if constexpr (PARK) {   // the next tile's K fragments are fetched once this tile's score MFMAs have issued (16 registers fewer in flight)
#pragma unroll
                    for (int ml = 0; ml < NML; ++ml) asm volatile("" : "+v"(s0[ml]), "+v"(s1[ml]));
                    ld_tile(kr, so0 + (unsigned)nx * 4096u, kn);
                }
                if (kt * 32 + 32 > Lk) {
#pragma unroll
                    for (int ml = 0; ml < NML; ++ml)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (kt * 32 + 4 * g + j >= Lk) s0[ml][j] = -INFINITY;
                            if (kt * 32 + 16 + 4 * g + j >= Lk) s1[ml][j] = -INFINITY;
                        }
                }
                float lm[NML];
                bool hot = kt == 0;
#pragma unroll
                for (int ml = 0; ml < NML; ++ml) {
                    lm[ml] = lane_max8(s0[ml], s1[ml], -INFINITY);
                    hot = hot || lm[ml] > CH_ATT_THR;
                }
                if (__builtin_amdgcn_ballot_w64(hot) != 0) {
                    // exact path (first tile; afterwards rare): the row maxima across the lane groups, m_run moves up to them, what was
                    // accumulated so far is rescaled and this tile's scores are shifted by the same amount
#pragma unroll
                    for (int ml = 0; ml < NML; ++ml) {
                        const float mx = ar4_max(lm[ml]);                       // relative to -nb
                        // a row moves its maximum on ITS OWN scores only (first tile, or past the threshold): what a row computes must
                        // not depend on which rows share its wave (a clip's sample does not depend on its batch)
                        const float m_new = (kt == 0 || mx > CH_ATT_THR) ? fmaxf(m_run[ml], mx - nb[ml]) : m_run[ml];
                        const float shift = m_new + nb[ml];
                        const float alpha = __builtin_amdgcn_exp2f(m_run[ml] - m_new);   // first tile: exp2(-inf) = 0, o and lacc are 0
                        m_run[ml] = m_new;
                        nb[ml] = -m_new;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            lacc[ml][j] *= alpha;
                            s0[ml][j] -= shift;
                            s1[ml][j] -= shift;
                        }
#pragma unroll
                        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[dt][ml][j] *= alpha;
                    }
                }
#pragma unroll
                for (int ml = 0; ml < NML; ++ml) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        s0[ml][j] = __builtin_amdgcn_exp2f(s0[ml][j]);
                        s1[ml][j] = __builtin_amdgcn_exp2f(s1[ml][j]);
                    }
                    u32x4 pf;
                    pf[0] = pack_bf2(s0[ml][0], s0[ml][1]);
                    pf[1] = pack_bf2(s0[ml][2], s0[ml][3]);
                    pf[2] = pack_bf2(s1[ml][0], s1[ml][1]);
                    pf[3] = pack_bf2(s1[ml][2], s1[ml][3]);
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) mma16(o[dt][ml], vc[dt], pf);
                    mma16(lacc[ml], ones, pf);
                }
            }
#pragma unroll
            for (int ml = 0; ml < NML; ++ml) {
                const float lsum = lacc[ml][0];
                if (my_seq[ml] == seq) {
                    const float inv = __builtin_amdgcn_rcpf(lsum);    // 1 ulp; the quotient is rounded to bf16 next
                    const int mt = NML * ps + ml;
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt) {
                        uint2 pk;
                        pk.x = pack_bf2(o[dt][ml][0] * inv, o[dt][ml][1] * inv);
                        pk.y = pack_bf2(o[dt][ml][2] * inv, o[dt][ml][3] * inv);
                        *reinterpret_cast<uint2*>(abuf + wave * 8192 + act_wr_off(lane, dt) + mt * 2048) = pk;
                    }
                }
            }
        }
    }
}

#ifdef CH_STAMP   // diagnostic build: per-phase timestamps of block 0, every wave, into the (otherwise unused) h_out buffer
#ifndef CH_STAMP_BLOCK
#define CH_STAMP_BLOCK 0      // the LOGICAL 64-row block that writes the stamps (7 = rows 448..511: straddles two 450-row sequences)
#endif
#define CH_T(i) do { if (m0 == 64 * CH_STAMP_BLOCK && (threadIdx.x & 63) == 0 && (MODE == TC_CHAIN_B || MODE == TC_CHAIN_FULL)) \
        reinterpret_cast<unsigned long long*>(a.h_out)[(threadIdx.x >> 6) * 64 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// the shader-clock counter beside the 100 MHz one: slots 60 / 61 = s_memtime at the first / last stamp
#define CH_TC(i) do { if (m0 == 64 * CH_STAMP_BLOCK && (threadIdx.x & 63) == 0 && (MODE == TC_CHAIN_B || MODE == TC_CHAIN_FULL)) \
        reinterpret_cast<unsigned long long*>(a.h_out)[(threadIdx.x >> 6) * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CH_TC(i) do { } while (0)
#define CH_T(i) do { } while (0)
#endif

// BRK: rows per block of the LAUNCH (16 x the kernel's MT); MT: row tiles this block computes -- BRK / 16, or 1 for a sequence's
// last block when it holds <= 16 rows (seq_blocks: 450 = 7 x 64 + 2; the block streams the weights like any other but runs a
// quarter of the MFMAs).
template <int MODE, int MT, int NT, int BRK>
DEVINL void chain_body(const tcdiff_chain_args& a) {
    constexpr int BR = BRK;            // rows per block (MT row tiles of them computed)
    constexpr int NW = ChW<NT>::NW, NTH = 64 * NW, RD = ChW<NT>::D;   // waves, threads, ring depth
    constexpr bool HAS_A = MODE == TC_CHAIN_A || MODE == TC_CHAIN_FULL || MODE == TC_CHAIN_FULL_LAST;   // fc + norm2 + w_qs
    constexpr bool FULL = MODE == TC_CHAIN_FULL || MODE == TC_CHAIN_FULL_LAST;                          // + cross-attention
    constexpr bool LAST = MODE == TC_CHAIN_B_LAST || MODE == TC_CHAIN_FULL_LAST;
    // TC_CHAIN_FRONT: the last fusion linear of ONE dancer for a block of 64 FRAMES (A = 64 rows of 1024), which is layer
    // 0's residual input, then layer 0's norm1 + rotary and Q / K / V -- the tail of chain B with a K = 1024 GEMM in
    // front.  Frame F, dancer d <-> token row F dn + d (model/model.py:561; model/diffusion.py:640,651).
    constexpr bool FRONT = MODE == TC_CHAIN_FRONT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int dn = FRONT ? a.dn : 1;
    const int dancer = FRONT ? (int)(blockIdx.x % (unsigned)dn) : 0;
    // logical block: FRONT (frame block, dancer); else consecutive logical blocks share an XCD (and its L2: FiLM rows, K / V).
    // seq_blocks: logical block = (sequence, block of the sequence); its rows end with the sequence (Mv), the clamped rows behind
    // them recompute the sequence's last row and are never stored.
    const int lblk = FRONT ? (int)(blockIdx.x / (unsigned)dn) : xcd_remap(blockIdx.x, gridDim.x);
    const bool seqcut = !FRONT && a.seq_blocks;
    const int nbs = seqcut ? (a.L + BR - 1) / BR : 1;                 // blocks per sequence
    const int bseq = seqcut ? lblk / nbs : 0, bis = seqcut ? lblk - bseq * nbs : 0;
    const int m0 = seqcut ? bseq * a.L + bis * BR : lblk * BR;
    CH_T(0);
    CH_TC(60);
    const int Mtot = a.M, L = a.L;     // FRONT: M = frames, L = TOKENS per sequence
    const int M = seqcut ? (bseq + 1) * L : Mtot;                     // first row past the block's valid rows
    char* abuf = smem + CH_ABUF;
    char* h1c = smem + CH_H1C;
    float* scr = reinterpret_cast<float*>(smem + CH_SCR);
    char* cfilm = smem + CH_FILM;      // [2 sequences][G: 512 | Bv: 512] floats
    char* cvec = smem + CH_VEC;        // six vectors of 512 floats

    // rows of this lane (four row tiles), clamped: rows past M recompute row M - 1 (their inputs are clamped to it).
    // Every phase recomputes what it needs of them from a fresh copy of the lane index (a few VALU ops) -- kept as arrays
    // they are 16 registers alive across the whole kernel.
    const int seq0 = (m0 < M ? m0 : M - 1) / L;
    const int seq_last = (Mtot - 1) / L;
    const int seqb = (seq0 + 1) * L;   // first row of the block's second sequence (L >= 64 rows per sequence: launcher)
    struct Rows { int mc[MT], sidx[MT]; };
    auto rows = [&]() {
        Rows r;
        const int cc = fresh_v(c);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = m0 + 16 * mt + cc;
            r.mc[mt] = m < M ? m : M - 1;
            r.sidx[mt] = r.mc[mt] >= seqb ? 1 : 0;
        }
        return r;
    };
    // rotary position of the rows: FRONT rows are frames of one dancer (token = frame dn + dancer)
    auto positions = [&](const Rows& r, int (&pos)[MT]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            pos[mt] = FRONT ? (r.mc[mt] * dn + dancer) % L : r.mc[mt] - (r.sidx[mt] ? seqb : seqb - L);
    };
    // Epilogue constants go through LDS.  Thread t carries one float4 of the FiLM rows and up to two of the vectors
    // from global memory to LDS; they are fetched early (latency hidden behind a GEMM) and stored once the previous
    // epilogue no longer reads the area.  The FiLM rows arrive PRE-FOLDED (tcdiff_chain_args.film): [G = g (scale + 1) |
    // Bv = b (scale + 1) + shift] with g, b the LayerNorm weights in front of the FiLM (or 1, linear2's bias), so an epilogue is
    // v = fma(u, G, Bv) + x (featurewise_affine, model/model.py:171-173, and the LayerNorm affine / bias in one step).
    // (512 float4 of FiLM rows and 768 of vectors over NTH threads: float4 index j NTH + tid)
    constexpr int CF = 512 / NTH, CV = (768 + NTH - 1) / NTH;
    struct Consts { f32x4_t f[CF], v[CV]; };
    auto fetch_consts = [&](const float* film, const float* const (&vec)[6]) {
        Consts k;
#pragma unroll
        for (int j = 0; j < CF; ++j) {
            const int idx = j * NTH + tid;
            int sq = seq0 + (idx >> 8);
            sq = sq < seq_last ? sq : seq_last;
            k.f[j] = film ? ld4(film + (long)sq * a.film_ld + (idx & 255) * 4) : f32x4_t{0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < CV; ++j) {
            const int idx = j * NTH + tid;
            // (a chain of selects: indexing the array with a lane-dependent index would put it in scratch)
            const int vi = idx >> 7;
            const float* p0 = vi == 0 ? vec[0] : vi == 1 ? vec[1] : vi == 2 ? vec[2] : vi == 3 ? vec[3] : vi == 4 ? vec[4] : vi == 5 ? vec[5] : nullptr;
            k.v[j] = p0 ? ld4(p0 + (idx & 127) * 4) : f32x4_t{0, 0, 0, 0};
        }
        return k;
    };
    auto store_consts = [&](const Consts& k) {
#pragma unroll
        for (int j = 0; j < CF; ++j) *reinterpret_cast<f32x4_t*>(cfilm + (j * NTH + tid) * 16) = k.f[j];
#pragma unroll
        for (int j = 0; j < CV; ++j)
            if (j * NTH + tid < 768) *reinterpret_cast<f32x4_t*>(cvec + (j * NTH + tid) * 16) = k.v[j];
    };
    auto vecp = [&](int slot) { return cvec + slot * 2048; };
    // constants of the fc block that opens chain B: its own set when chain A ran in front of it in this launch
    const float* fcb_film = FULL ? a.filmb : a.film;
    const float* n3_g = FULL ? a.n3_g : a.n2_g;
    const float* n3_b = FULL ? a.n3_b : a.n2_b;

    // ---- the block's input rows (attention output) -> LDS, the first CH_D weight stages -> registers, constants -> LDS
    // (the LDS image keeps the 64-row geometry -- 8 KB per k-tile -- whatever MT: rows 16 MT .. 63 are simply unused)
    constexpr int NWL = 2 * MT < NW ? 2 * MT : NW;     // waves that stage the rows (8 rows per wave instruction)
    // sa_q: the rows are not read but computed -- this layer's self-attention runs here, from the fragments the previous launch left
    const bool self_att = FULL && NT == 4 && a.sa_q != nullptr;
    WStreamT<NT> ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.wstream)) + ((long)dancer * NW + wave) * a.n_stages * ChW<NT>::STAGE, 0,
        a.n_stages * ChW<NT>::STAGE, 0x00020000);   // raw buffer (stride 0), bounds = the wave's stream, 32-bit data format
    ws.voff = (unsigned)lane * 16u;
    ws.pos = 0;
    ws.last = (unsigned)a.n_stages - 1;
    auto ring_fill = [&]() {
#pragma unroll
        for (int i = 0; i < RD; ++i) ws_load(ws, i, (unsigned)i);
    };
    f32x4_t acc[NT][MT];
    if constexpr (FULL && NT == 4) {
        if (self_att) {
            // ============= self-attention of the block's rows (model/model.py:97-102,326-327): O -> the activation block (wave-private
            // columns; the barrier below publishes them).  The weight ring is filled behind it: its 64 registers are the attention's.
            cross_attention<0, MT, NT, true>(acc, a, m0, M, lblk, abuf, wave, lane);
            ring_fill();      // (issued from inside the attention's epilogue the ring spills: its 64 registers beside the 64 of O)
        }
    }
    if (!self_att) {
        if (wave < NWL) {
#pragma unroll
            for (int kt = 0; kt < (FRONT ? 16 : 8); ++kt)  // FRONT: rows of 1024 = the activation block and its twin, contiguous
                stage_glds<16 * MT, NWL>(abuf + kt * 8192, reinterpret_cast<const char*>(a.A) + kt * TC_ROWB, FRONT ? 2048 : 1024, m0,
                                    M, a.a_mod, wave, lane);
        }
        ring_fill();
    }
    if (FRONT) {
        const float* const v[6] = {a.b3 + 512 * dancer, a.nn_g, a.nn_b, nullptr, nullptr, nullptr};
        store_consts(fetch_consts(nullptr, v));
    } else {
        const float* const v[6] = {nullptr, nullptr, a.n2_g, a.n2_b, nullptr, nullptr};
        store_consts(fetch_consts(a.film, v));
    }
    // a wait the compiler can SEE (an asm s_waitcnt is invisible to its vmcnt bookkeeping: it would then treat the
    // prologue loads as still pending at the loop header and wait vmcnt(0) on every trip): vmcnt(0), others untouched
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    CH_T(1);


    float nmr[MT], rstd[MT];             // LayerNorm of the current rows: u = fma(v, rstd, nmr)
    RowPipe<MT> rp;                      // residual rows, later rotary rows, of this lane
    const long xrows = (long)Mtot * dn;
    const __amdgpu_buffer_rsrc_t xo = f32_buffer(a.xout, xrows * 512);     // the residual stream out
    Consts nxt;

    // fc epilogue: LayerNorm(eps), FiLM, residual -> x in the accumulators and in xout (model/model.py:103-106,171-173,
    // 327 / 334); constants in vector slots 0, 1 and the FiLM area; the residual rows were started by the caller.  The
    // barrier inside the statistics exchange is also the one that says every wave has left the GEMM.
    auto fc_epilogue = [&](float eps, int stamp) {
        CH_T(stamp);
        row_stats(acc, scr, wave, lane, eps, nmr, rstd);
        CH_T(stamp + 1);
        // fresh copies: the two inlined instances of this epilogue must not share (and keep alive) their addresses
        const int gg = fresh_v(g), wv = fresh_s(wave);
        const Rows rw = rows();
        const int cb0 = col_base_bytes<NT>(wv, gg);
        auto body = [&](int nt, int mt, const f32x4_t& G, const f32x4_t& Bv) {
            const f32x4_t x4 = rp.q[nt & 1][mt];
            f32x4_t o;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v = fmaf(fmaf(acc[nt][mt][t], rstd[mt], nmr[mt]), G[t], Bv[t]);
                v = x4[t] + v;
                acc[nt][mt][t] = v;
                o[t] = v;
            }
#ifdef CH_STORE_EARLY   // (rounds 2-5: the store beside the arithmetic; A/B build flag)
            cb_store<NT>(xo, Mtot, wv, nt, rw.mc[mt], gg, o);
#endif
        };
        // (Measured in the listing and dropped: a second code path for blocks that lie in ONE sequence -- constants read once
        // per column quad instead of once per row tile.  The two paths raise the epilogue's register peak, hipcc spills ring
        // slots across it, and the launch ends with its straddling blocks anyway.)
        int fb[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            fb[mt] = rw.sidx[mt] * 4096 + cb0;    // FiLM rows of this lane's rows: sequence 0 or 1 of the block (4 KB apart)
            asm volatile("" : "+v"(fb[mt]));
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                body(nt, mt, lds4b(cfilm + fb[mt], 64 * nt), lds4b(cfilm + fb[mt], 2048 + 64 * nt));
            if (nt + 2 < NT) CH_RP_NEXT(rp, nt + 2);
            // one n-tile at a time: without a fence hipcc hoists the loads of ALL of them (row pipeline refills and LDS
            // constants) above the arithmetic, needs ~100 more registers and spills them
            CH_FENCE();
        }
#ifndef CH_STORE_EARLY
        // The new x leaves AFTER the loop, from the accumulators (they keep it for the next norm anyway).  gfx950 counts loads and
        // stores in ONE in-order counter (vmcnt): a store issued in front of the next n-tile's row loads made the wait for those
        // loads a wait for the store's acknowledgement too -- 1-2 us under load, twice per epilogue (round 6: the ISA showed
        // `store, load, s_waitcnt vmcnt(2)` chains; the epilogue took 2.6 us per wave for 0.6 us of arithmetic).
        // unguarded: rows past M rewrite row M - 1 with the SAME values (same inputs; every load of this column group, in place
        // or not, was issued before these stores)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) cb_store<NT>(xo, Mtot, wv, nt, rw.mc[mt], gg, acc[nt][mt]);
#endif
    };
    auto xres_start = [&]() {
        const Rows rw = rows();
        int rr[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) rr[mt] = a.xres_mod > 0 ? rw.mc[mt] % a.xres_mod : rw.mc[mt];
        rp_start<NT>(rp, a.xres, rr, a.xres_rowmajor ? 0 : (a.xres_mod > 0 ? a.xres_mod : Mtot), a.xres_mod > 0 ? a.xres_mod : Mtot, wave,
                     fresh_v(g));
    };
    auto xout_start = [&]() {          // the x this lane stored in an earlier epilogue of this launch
        const Rows rw = rows();
        rp_start<NT>(rp, a.xout, rw.mc, Mtot, Mtot, wave, fresh_v(g));
    };
    auto rope_start = [&]() {
        const Rows rw = rows();
        int pos[MT];
        positions(rw, pos);
        rp_start<NT>(rp, a.rope, pos, a.rope_rows, a.rope_rows, wave, fresh_v(g));
    };

    if (HAS_A) {
        // ================= self-attention block tail: fc + LayerNorm(1e-6) + FiLM + residual, norm2 + rotary, w_qs
        zero(acc);
        phase_n512<16, false, MT, NT>(acc, abuf, ws, lane);
        CH_T(2);
        xres_start();                  // in flight during the statistics exchange
        fc_epilogue(a.ln_eps, 40);
        CH_T(3);
        if (FULL) {
            const float* const v[6] = {nullptr, nullptr, n3_g, n3_b, nullptr, nullptr};
            nxt = fetch_consts(fcb_film, v);
        }
        rope_start();
        row_stats(acc, scr + 1024, wave, lane, a.n2_eps, nmr, rstd);
        CH_T(4);
        // norm2 + rotary (model/model.py:332,387) -> LDS -> Q = rot W_q^T / 8 (model/model.py:78,97)
        norm_to_lds<true, MT, NT>(acc, nmr, rstd, vecp(2), vecp(3), rp, abuf, wave, lane, nullptr);
        lds_barrier();
        CH_T(34);
        if (FULL) store_consts(nxt);   // the cross-attention fc block's constants: read two barriers from here
        zero(acc);
        if (!FULL) {
            phase_n512<16, false, MT, NT>(acc, abuf, ws, lane);
            store_heads<true, MT, NT>(acc, a.q_out, a.scale_q, L, a.Lp, a.H, m0, M, wave, lane, smem);
            return;
        }
#ifdef CH_STAMP   // per-stage shader-clock stamps of the w_qs GEMM: h_out[512 + wave * 32 + stage], [.. + 16] = the phase's start
        {
            unsigned long long* sp = nullptr;
            if (m0 == 64 * CH_STAMP_BLOCK) {
                sp = reinterpret_cast<unsigned long long*>(a.h_out) + 512 + wave * 32;
                if (lane == 0) sp[16] = __builtin_amdgcn_s_memtime();
            }
            phase_n512<16, false, MT, NT>(acc, abuf, ws, lane, sp);
        }
#else
        phase_n512<16, false, MT, NT>(acc, abuf, ws, lane);
#endif
        // ================= cross-attention in place (the Q image never leaves the registers)
        CH_T(35);
        lds_barrier();                 // every wave is out of the w_qs GEMM: the activation block becomes O
#ifndef CH_ABLATE_XATTN   // (timing experiment)
        cross_attention<0>(acc, a, m0, M, lblk, abuf, wave, lane);
        if constexpr (NT == 8) cross_attention<1>(acc, a, m0, M, lblk, abuf, wave, lane);
#endif
        CH_T(36);
        lds_barrier();
    }
    if constexpr (!FRONT) {
    // ================= cross-attention block tail: fc + LayerNorm(1e-6) + FiLM + residual (model/model.py:334)
    zero(acc);
    phase_n512<16, false, MT, NT>(acc, abuf, ws, lane);
    if (FULL)
        xout_start();                  // the x this lane stored in the first fc epilogue
    else
        xres_start();
    CH_T(37);
    fc_epilogue(a.ln_eps, 42);
    CH_T(38);
    row_stats(acc, scr + 1024, wave, lane, a.n2_eps, nmr, rstd);
    CH_T(39);
    // ================= feed-forward (model/model.py:338-339,399-401): norm3 -> LDS
    {
        const float* const v[6] = {a.b1, a.b1 + 512, nullptr, a.n4_g, a.n4_b, nullptr};
        nxt = fetch_consts(a.film3, v);
    }
    norm_to_lds<false, MT, NT>(acc, nmr, rstd, vecp(2), vecp(3), rp, abuf, wave, lane, nullptr);
    lds_barrier();                     // nobody reads the fc constants any more
    store_consts(nxt);
    lds_barrier();                     // ... and everybody sees the feed-forward constants
    CH_T(5);
    zero(acc);   // acc = linear2 accumulator
    // (Measured and dropped, round 4: the chunks as a software pipeline -- linear1(c), then linear2(c - 1) with the GELU of
    // chunk c issued one tile per stage behind that stage's MFMAs, the weight stream packed in that order.  Same cycles per
    // launch (208.8 k against 207.9 k shader cycles at 225 blocks): the SIMD's two waves already run half a phase apart --
    // the older one takes the matrix pipe first -- so the leader's GELU sits under the follower's MFMAs as it is, and the
    // fused phase is bound by VALU issue (~300 cycles of erf polynomial per 16 x 16 tile, two waves) instead.)
#pragma unroll 1
    for (int ch = 0; ch < 4; ++ch) {
        f32x4_t a1[NT / 2][MT];
#pragma unroll
        for (int nt = 0; nt < NT / 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) a1[nt][mt] = f32x4_t{0, 0, 0, 0};
        phase_ff1<MT, NT>(a1, abuf, ws, lane);
        CH_T(6 + 4 * ch);
        // two h1 buffers: chunk c - 2's linear2 reads of this one finished before the barrier of chunk c - 1
        char* hb = h1c + (ch & 1) * 32768;
        CH_T(7 + 4 * ch);
        {
            const int nb = 256 * ch + 8 * NT * wave + 4 * g;    // b1 occupies vector slots 0 and 1
#pragma unroll
            for (int nt = 0; nt < NT / 2; ++nt) {
                const f32x4_t b4 = lds4(cvec, nb + 16 * nt);
                // chunk column cc = 8 NT wave + 16 nt (+ 4 g): k-tile cc / 64, 16-byte chunk (cc % 64) / 8 + (g >> 1)
                const int wo = ((8 * NT * wave + 16 * nt) >> 6) * 8192 + act_wr_off(lane, 0, ((8 * NT * wave + 16 * nt) & 63) >> 3);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    float v[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = a1[nt][mt][t] + b4[t];
#ifndef CH_ABLATE_GELU   // (timing experiment: how much of the launch is the GELU's VALU work?)
                    act4_ct<ACT_GELU>(v, ACT_GELU);
#endif
                    uint2 pk;
                    pk.x = pack_bf2(v[0], v[1]);
                    pk.y = pack_bf2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(hb + wo + mt * 2048) = pk;
                }
            }
        }
        lds_barrier();
        CH_T(8 + 4 * ch);
        phase_n512<8, false, MT, NT>(acc, hb, ws, lane);
        CH_T(9 + 4 * ch);
    }
    // linear2 bias, FiLM, residual (the x this lane stored above), norm4 -> LDS
    xout_start();
    {
        const int cb2 = col_base_bytes<NT>(fresh_s(wave), fresh_v(g));
        const Rows rw = rows();
        auto body2 = [&](int nt, int mt, const f32x4_t& G, const f32x4_t& Bv) {
            const f32x4_t x4 = rp.q[nt & 1][mt];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[nt][mt][t] = x4[t] + fmaf(acc[nt][mt][t], G[t], Bv[t]);     // linear2's bias is in Bv
            // nothing in this iteration touches memory after its loads, so the arithmetic is free to sink below the
            // loads of all later iterations (whose operands then all have to be kept): pin it to this iteration
            asm volatile("" ::"v"(acc[nt][mt][0]), "v"(acc[nt][mt][1]), "v"(acc[nt][mt][2]), "v"(acc[nt][mt][3]));
        };
        int fb2[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            fb2[mt] = rw.sidx[mt] * 4096 + cb2;
            asm volatile("" : "+v"(fb2[mt]));
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                body2(nt, mt, lds4b(cfilm + fb2[mt], 64 * nt), lds4b(cfilm + fb2[mt], 2048 + 64 * nt));
            if (nt + 2 < NT) CH_RP_NEXT(rp, nt + 2);
            CH_FENCE();
        }
    }
    CH_T(22);
    {
        const float* const v[6] = {a.b3, a.nn_g, a.nn_b, nullptr, nullptr, nullptr};
        nxt = fetch_consts(nullptr, v);
    }
    row_stats(acc, scr, wave, lane, a.n4_eps, nmr, rstd);      // (its barrier: every wave is out of the last linear2 chunk)
    CH_T(23);
    norm_to_lds<false, MT, NT>(acc, nmr, rstd, vecp(3), vecp(4), rp, abuf, wave, lane, nullptr);
    lds_barrier();
    store_consts(nxt);                 // b3, norm1': read after the barrier that follows linear3
    }   // !FRONT
    // ================= x' = linear3(norm4(x)) + b3, no residual (model/model.py:344); FRONT: the last fusion linear of this
    // block's dancer over K = 1024 (model/model.py:526-528), whose output is layer 0's residual input
    CH_T(24);
    zero(acc);
    if (FRONT)
        phase_n512<32, false, MT, NT>(acc, abuf, ws, lane);
    else if (LAST)
        phase_n512<16, true, MT, NT>(acc, abuf, ws, lane);
    else
        phase_n512<16, false, MT, NT>(acc, abuf, ws, lane);
    CH_T(25);
    lds_barrier();
#ifndef CH_STORE_EARLY
    if (!LAST) rope_start();           // the next norm's rotary rows: issued BEFORE x' is stored (one in-order vmcnt: behind the stores,
                                       // waiting for these loads would wait for the stores' acknowledgement too -- see fc_epilogue)
#endif
    {
        const int g3 = fresh_v(g);
        const int cb3 = col_base_bytes<NT>(fresh_s(wave), g3);
        const Rows rw = rows();
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = 16 * NT * wave + 16 * nt + 4 * g3;
            const f32x4_t b4 = lds4b(vecp(0) + cb3, 64 * nt);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x4_t o;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[nt][mt][t] += b4[t];
                    o[t] = acc[nt][mt][t];
                }
                if (LAST && a.out_ld > 0) {
                    // linear3 carries the final projection folded into it (engine.py: W_final W_3, two linear maps
                    // with nothing in between, model/model.py:344,623): columns [0, out_ld) ARE the network output
                    if (n < a.out_ld)
                        *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(a.h_out) + (long)rw.mc[mt] * a.out_ld + n) = o;
                } else if (LAST) {   // a separate final projection reads bf16 rows (model/model.py:623)
                    uint2 pk;
                    pk.x = pack_bf2(o[0], o[1]);
                    pk.y = pack_bf2(o[2], o[3]);
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.h_out) + (long)rw.mc[mt] * 512 + n) = pk;
                } else {
                    cb_store<NT>(xo, xrows, wave, nt, rw.mc[mt] * dn + dancer, g3, o);
                }
            }
            CH_FENCE();
        }
    }
    if (LAST) return;
    // ================= next layer: norm1 + rotary -> Q, K ; norm1 -> V (model/model.py:326,374-383,78-80)
    CH_T(26);
#ifdef CH_STORE_EARLY
    rope_start();
#endif
    row_stats(acc, scr, wave, lane, a.nn_eps, nmr, rstd);
    CH_T(27);
    norm_to_lds<true, MT, NT>(acc, nmr, rstd, vecp(1), vecp(2), rp, abuf, wave, lane, smem + CH_ABUF2);
    lds_barrier();
    CH_T(28);
    zero(acc);
    phase_n512<16, false, MT, NT>(acc, abuf, ws, lane);
    CH_T(29);
    // qf_out: Q / K / V leave in the fragment order of the next launch's in-kernel self-attention (seq_blocks; 8-wave form)
    constexpr bool FRAG = !FRONT && NT == 4;
    bool frag_out = false;
    if constexpr (FRAG) frag_out = a.qf_out != nullptr;
    if constexpr (FRAG) {
        if (frag_out) store_qfrag<MT>(acc, a.qf_out, a.scale_q, lblk, wave, lane);
    }
    if (!frag_out) store_heads<true, MT, NT>(acc, a.q_out, a.scale_q, L, a.Lp, a.H, m0, M, wave, lane, smem, dn, dancer);
    zero(acc);
    phase_n512<16, false, MT, NT>(acc, abuf, ws, lane);
    CH_T(30);
    if constexpr (FRAG) {
        if (frag_out) store_kfrag<MT>(acc, a.kf_out, bseq, bis * BR, a.out_nkt, wave, lane);
    }
    if (!frag_out) store_heads<false, MT, NT>(acc, a.k_out, 1.0f, L, a.Lp, a.H, m0, M, wave, lane, smem, dn, dancer);
    CH_T(31);
    zero(acc);
    if constexpr (FRAG) {
        if (frag_out) {
            phase_n512<16, true, MT, NT, true>(acc, smem + CH_ABUF2, ws, lane);
            CH_T(32);
            store_vfrag<MT>(acc, a.vf_out, bseq, bis * BR, a.out_nkt, L, wave, lane);
            CH_T(33);
            CH_TC(61);
            return;
        }
    }
    phase_n512<16, true, MT, NT>(acc, smem + CH_ABUF2, ws, lane);
    CH_T(32);
    store_heads<false, MT, NT>(acc, a.v_out, 1.0f, L, a.Lp, a.H, m0, M, wave, lane, smem, dn, dancer);
    CH_T(33);
    CH_TC(61);
}

// MT: 16-row tiles per block (4: 64-row blocks; 2, 1: small jobs, see tcdiff_chain)
template <int MODE, int MT, int NT>
__global__ __launch_bounds__(2048 / NT) void chain_kernel(tcdiff_chain_args a) {
    constexpr bool DUAL = (MODE == TC_CHAIN_FULL || MODE == TC_CHAIN_FULL_LAST) && NT == 4 && MT > 1;
    if constexpr (DUAL) {
        if (a.seq_blocks) {      // a sequence's last block with <= 16 rows computes one row tile
            const int lblk = xcd_remap(blockIdx.x, gridDim.x);
            const int nbs = (a.L + 16 * MT - 1) / (16 * MT);
            if (a.L - (lblk % nbs) * 16 * MT <= 16) {
                chain_body<MODE, 1, NT, 16 * MT>(a);
                return;
            }
        }
    }
    chain_body<MODE, MT, NT, 16 * MT>(a);
}

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int tcdiff_chain(const tcdiff_chain_args* a, hipStream_t stream) {
    if (!a || a->M <= 0 || a->L <= 0 || !a->A || !a->wstream) return TC_ERR_ARG;
    const bool front = a->mode == TC_CHAIN_FRONT;
    if (!front && a->L < 64) return TC_ERR_UNSUPPORTED;   // a 64-row block must touch at most two sequences
    const int want = a->mode == TC_CHAIN_A ? 32 : a->mode == TC_CHAIN_B ? 144 : a->mode == TC_CHAIN_B_LAST ? 96 :
                     a->mode == TC_CHAIN_FULL ? 176 : a->mode == TC_CHAIN_FULL_LAST ? 128 : front ? 80 : -1;
    if (want < 0 || a->n_stages != want) return TC_ERR_ARG;
    if (a->out_ld < 0 || a->out_ld % 4 || a->out_ld > 512) return TC_ERR_ARG;
    const void* ptrs[] = {a->A, a->wstream, a->film, a->xres, a->xout, a->n2_g, a->n2_b, a->rope,
                          a->q_out, a->b1, a->film3, a->n4_g, a->n4_b, a->b3, a->nn_g, a->nn_b, a->k_out,
                          a->v_out, a->h_out, a->filmb, a->n3_g, a->n3_b, a->kf, a->vf};
    for (const void* p : ptrs)
        if (p && !al16(p)) return TC_ERR_ALIGN;
    if (front) {
        // A = bf16 [M frames][1024]; b3 = the 512 dn biases of the last fusion linear; xout = layer 0's residual input
        if (a->dn <= 0 || a->L % a->dn || a->L / a->dn < 8 || !a->b3 || !a->nn_g || !a->nn_b || !a->rope || !a->xout ||
            !a->q_out || !a->k_out || !a->v_out || a->H != 8 || a->Lp <= 0 || a->a_mod != 0)
            return TC_ERR_ARG;
        if ((long)a->M * 2048 >= (1L << 32)) return TC_ERR_ARG;
    } else {
        if (!a->film || a->film_ld % 4 || !a->xres || !a->xout || !a->n2_g || !a->n2_b) return TC_ERR_ARG;
        if ((long)(a->a_mod > 0 ? a->a_mod : a->M) * 1024 >= (1L << 32)) return TC_ERR_ARG;
    }
    const bool has_a = a->mode == TC_CHAIN_A || a->mode == TC_CHAIN_FULL || a->mode == TC_CHAIN_FULL_LAST;
    const bool has_b = a->mode != TC_CHAIN_A && !front;
    const bool full = a->mode == TC_CHAIN_FULL || a->mode == TC_CHAIN_FULL_LAST;
    const bool last = a->mode == TC_CHAIN_B_LAST || a->mode == TC_CHAIN_FULL_LAST;
    if (has_a && (!a->rope || a->H != 8)) return TC_ERR_ARG;
    if (a->mode == TC_CHAIN_A && (!a->q_out || a->Lp <= 0)) return TC_ERR_ARG;
    if (has_b && (!a->b1 || !a->film3 || !a->n4_g || !a->n4_b || !a->b3)) return TC_ERR_ARG;
    const bool frag_out = a->qf_out || a->kf_out || a->vf_out;      // fragment-order Q / K / V instead of the head-major images
    if (has_b && !last &&
        (!a->rope || !a->nn_g || !a->nn_b || (!frag_out && (!a->q_out || !a->k_out || !a->v_out)) || a->H != 8 || a->Lp <= 0))
        return TC_ERR_ARG;
    if (last && !a->h_out) return TC_ERR_ARG;
    if (full && (!a->filmb || !a->n3_g || !a->n3_b || !a->kf || !a->vf || a->nkt <= 0 ||
                 a->Lk <= 0 || a->Lk > 32 * a->nkt || a->n_shared < 0))
        return TC_ERR_ARG;
    // sequence-cut blocks, the in-kernel self-attention and the fragment-order outputs (round 5)
    if (a->seq_blocks && (front || a->M % a->L)) return TC_ERR_ARG;
    if (a->sa_q && (!full || !a->seq_blocks || !a->sa_kf || !a->sa_vf || a->sa_nkt <= 0 || a->L > 32 * a->sa_nkt)) return TC_ERR_ARG;
    if (frag_out && (!a->qf_out || !a->kf_out || !a->vf_out || !has_b || last || !a->seq_blocks || a->out_nkt <= 0 ||
                     a->L > 32 * a->out_nkt))
        return TC_ERR_ARG;
    if ((a->sa_q || frag_out) && a->nw == 4) return TC_ERR_UNSUPPORTED;
    for (const void* p : {a->sa_q, a->sa_kf, a->sa_vf, (const void*)a->qf_out, (const void*)a->kf_out, (const void*)a->vf_out})
        if (p && !al16(p)) return TC_ERR_ALIGN;
    if (a->sa_q && (long)(a->M / a->L) * 8 * a->sa_nkt * 4096 >= (1L << 32)) return TC_ERR_ARG;
    static tc_dev_state dev_state;
    const int n_cu = tc_device_once(dev_state, [](int) {
#define CH_FN(M_, NT_) reinterpret_cast<const void*>(chain_kernel<M_, 4, NT_>), reinterpret_cast<const void*>(chain_kernel<M_, 2, NT_>), \
                        reinterpret_cast<const void*>(chain_kernel<M_, 1, NT_>)
        const void* fns[27] = {CH_FN(TC_CHAIN_A, 4), CH_FN(TC_CHAIN_B, 4), CH_FN(TC_CHAIN_B_LAST, 4), CH_FN(TC_CHAIN_FULL, 4),
                               CH_FN(TC_CHAIN_FULL_LAST, 4), CH_FN(TC_CHAIN_FRONT, 4),
                               CH_FN(TC_CHAIN_FULL, 8), CH_FN(TC_CHAIN_FULL_LAST, 8), CH_FN(TC_CHAIN_FRONT, 8)};
#undef CH_FN
        for (const void* f : fns) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, CH_SMEM);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    });
    if (n_cu < 0) return n_cu;
    // Rows per block: 64 when that gives the chip enough blocks, else 32 or 16 -- a block streams the whole layer's weights
    // whatever its rows (>= 48 us at the ~115 GB/s a CU takes in), so a small job (a few clips: what TCDiff.py renders,
    // TCDiff.py:292-303) is fastest on MANY small blocks, one per CU; results do not depend on the choice beyond fp32
    // summation order (a->mt forces it: tests).
    const int units = front ? a->dn : 1;
    int mt = a->mt;
    if (mt == 0) mt = ((a->M + 15) / 16) * units <= n_cu ? 1 : ((a->M + 31) / 32) * units <= n_cu ? 2 : 4;
    if (mt != 1 && mt != 2 && mt != 4) return TC_ERR_ARG;
    dim3 grid(((a->M + 16 * mt - 1) / (16 * mt)) * units);
    if (a->seq_blocks) grid.x = (a->M / a->L) * ((a->L + 16 * mt - 1) / (16 * mt));
    // a->nw: waves per workgroup.  0 / 8: eight waves of 64 columns (two per SIMD); 4: four waves of 128 columns, one per SIMD
    // with the 512-register budget (the production modes only: FULL, FULL_LAST, FRONT; the weight stream is packed per form)
    if (a->nw != 0 && a->nw != 4 && a->nw != 8) return TC_ERR_ARG;
    const bool four = a->nw == 4;
    if (four && !(full || front)) return TC_ERR_UNSUPPORTED;
#define CH_LAUNCH_NT(MODE_, NT_)                                                                                              \
    do {                                                                                                                      \
        if (mt == 4) hipLaunchKernelGGL((chain_kernel<MODE_, 4, NT_>), grid, dim3(2048 / NT_), CH_SMEM, stream, *a);          \
        else if (mt == 2) hipLaunchKernelGGL((chain_kernel<MODE_, 2, NT_>), grid, dim3(2048 / NT_), CH_SMEM, stream, *a);     \
        else hipLaunchKernelGGL((chain_kernel<MODE_, 1, NT_>), grid, dim3(2048 / NT_), CH_SMEM, stream, *a);                  \
    } while (0)
#define CH_LAUNCH(MODE_) CH_LAUNCH_NT(MODE_, 4)
#define CH_LAUNCH2(MODE_) do { if (four) CH_LAUNCH_NT(MODE_, 8); else CH_LAUNCH_NT(MODE_, 4); } while (0)
    switch (a->mode) {
        case TC_CHAIN_FRONT: CH_LAUNCH2(TC_CHAIN_FRONT); break;
        case TC_CHAIN_A: CH_LAUNCH(TC_CHAIN_A); break;
        case TC_CHAIN_B: CH_LAUNCH(TC_CHAIN_B); break;
        case TC_CHAIN_B_LAST: CH_LAUNCH(TC_CHAIN_B_LAST); break;
        case TC_CHAIN_FULL: CH_LAUNCH2(TC_CHAIN_FULL); break;
        default: CH_LAUNCH2(TC_CHAIN_FULL_LAST); break;
    }
#undef CH_LAUNCH
#undef CH_LAUNCH2
#undef CH_LAUNCH_NT
    TC_CHECK_LAUNCH();
    return TC_OK;
}
