// MFMA GEMMs of the TCDiff denoiser: C[M,N] = A[M,K] * W[N,K]^T (+ fused epilogues), gfx950.
//
// A is an activation matrix (row-major, K contiguous), W is an nn.Linear weight as stored by torch
// ([out, in] = [N, K], K contiguous), so both operands stage as rows of K and every MFMA fragment is a
// single 16-byte LDS read (common.h).  Two kernels:
//
//   gemm_tile   128x128 block tile, 4 waves (2x2), 2 workgroups per CU.  Replaces the aten addmm/linear calls of
//               model/model.py:78-80 (w_qs/w_ks/w_vs), :399 (linear1), :522-528 (fusion projection),
//               :560 (input_projection), :623 (final_layer), :454-465 (time MLP), :164-166 (FiLM).
//   gemm_rowln  64x512 block tile, 8 waves (2x4): one workgroup owns complete 512-wide rows, so the
//               LayerNorm / FiLM / residual that follow fc, linear2 and linear3 in the reference
//               (model/model.py:103-106,327,334,339,344) run in the epilogue, and the NEXT op's
//               LayerNorm(+rotary) input is emitted too (model/model.py:326,332,338,375,387).  No intermediate
//               round-trips HBM between a GEMM and the norm that follows it.
//
// Staging: global -> LDS directly with global_load_lds_dwordx4 (no VGPR staging, no ds_write pass).  One
// wave-instruction lands 64 lanes x 16 B = 8 tile rows linearly in LDS; the XOR swizzle of common.h::tile_off
// is applied on the per-lane SOURCE address (cdna_hip_programming.md rule 21).  Two LDS stages; the barrier that
// closes a k-step is preceded by an explicit s_waitcnt vmcnt(0) (common.h sync_dma) that retires the next tile's DMA.
// Measured (tools/microbench*.py, tools/dma/): one global_load_lds costs the issuing wave ~200 cycles, and a CU pulls
// ~55-60 GB/s from L2 into LDS at this tile shape whether by LDS-DMA or by register staging: that rate, not the MFMA
// pipe, bounds the main loop at B = 16 (56 rows of activations per CU against all the weights).
// Two restructurings were built and measured against this kernel and dropped (profiles/README.md):
//   * a persistent 256x128 kernel with dedicated loader waves and a 3-slot ring across tiles: with 4 loader waves a
//     48 KB stage took 1.4 us, with 8 loader waves 1.0 us, of which 0.72 us is the burst in which the eight waves issue
//     their 48 DMA instructions: ~67 GB/s is the rate at which a CU takes LDS-DMA instructions whoever issues them
//     (a wave alone needs ~120 cycles per instruction, eight together ~230 each); end to end 5 % slower than this kernel;
//   * one near-square tile per CU (16 waves, 225..343 rows x 256 columns, half the staged bytes, separate A / W rings):
//     with the DMA off a K step took 80 % of the MFMA rate at the 1.72 GHz the clock drops to, with it on 61 %; end to
//     end within 3 % of this kernel on the step's three shapes, because prologue + epilogue are not overlapped there.
//
// Epilogue stores go through LDS so that global writes are whole 16-byte chunks of contiguous rows:
//   gemm_tile issues the MFMAs with the operand roles swapped (A operand = weight rows, B operand = activation
//   rows), so each lane ends up holding 4 CONSECUTIVE output columns of one output row per register quad;
//   gemm_rowln dumps its fp32 64x512 tile to LDS and finishes LayerNorm/FiLM/residual/rotary in a row-wise pass
//   (one wave per row, 16-byte loads/stores, wave-wide shuffle reductions) exactly like ops.hip::ln_rot.
#include "attn_common.h"      // v_frag: transposed MFMA fragments of a staged [k][64 columns] tile (the TN form)
#include "train_common.h"     // counter-hash dropout, activation derivatives (the fused activation epilogues of the training step)
#include "tcdiff_hip.h"

#ifdef TC_STAMP
// diagnostic build (-DTC_STAMP, tools/microbench3.py): per-block timestamps (100 MHz realtime counter) written to a
// side buffer that nothing else reads; never compiled into the product library.
__device__ unsigned long long* g_tc_stamp = nullptr;
extern "C" int tcdiff_debug_stamp_buffer(void* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tc_stamp), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define TC_STAMP_AT(i) do { if (tid == 0 && g_tc_stamp) g_tc_stamp[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define TC_STAMP_W(i) do { if (lane == 0 && g_tc_stamp && (i) < 32) g_tc_stamp[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TC_STAMP_AT(i) do { } while (0)
#define TC_STAMP_W(i) do { } while (0)
#endif

// =================================================================================================
// gemm_tile: 128 x 128 x (128 B of K) tiles, 256 threads
// =================================================================================================
// NS = LDS stages of 32 KB ([A tile | W tile] of one k-tile).  Two stages (64 KB, two workgroups per CU) when the grid
// has tiles to spare; FOUR (128 KB, one workgroup per CU) when it has about one tile per CU -- the per-step GEMMs
// outside the decoder layers (FiLM stack, input / fusion / final projection: 152-228 tiles, K up to 1536).  With two
// stages every k-tile waits out what is left of a ~1.2 us DMA round trip after 0.25 us of MFMA work (25 us for K =
// 1536); with four the DMA of tile kt + 3 is issued while tile kt is computed.
// a 16-byte chunk of T <-> floats
template <class P>
DEVINL void unpack_chunk(const u32x4& c, float (&v)[16 / sizeof(typename P::elem_t)]) {
    if (P::IS_BF16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[2 * j] = bf2f((uint16_t)(c[j] & 0xffffu));
            v[2 * j + 1] = bf2f((uint16_t)(c[j] >> 16));
        }
    } else {
        const f32x4_t f = P::chunk_to4(c);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = f[j];
    }
}
template <class P>
DEVINL u32x4 pack_chunk(const float (&v)[16 / sizeof(typename P::elem_t)]) {
    u32x4 c;
    if (P::IS_BF16) {
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = pack_bf2(v[2 * j], v[2 * j + 1]);
    } else {
        const f32x4_t f = {v[0], v[1], v[2], v[3]};
        c = P::chunk_from4(f);
    }
    return c;
}

// TN = true: out[m][n] = sum_k A[k][m] W[k][n] -- BOTH operands row-major over the contraction index (the weight gradient
// dW = dY^T X straight from the token-major dY and X: no transposed copies).  A k-tile is staged as [KT rows of k][128
// columns] in 128-byte column blocks (the layout of attention's V tile), and every fragment is the transposed read of
// attn_common.h::v_frag (ds_read_b64_tr_b16 for bf16).  Requires M, N multiples of 128 and K a multiple of the k-tile.
template <class P, int ACT, int NS, bool TN = false>
__global__ __launch_bounds__(256) void gemm_tile_kernel(const char* __restrict__ A, const char* __restrict__ A2,
                                                        int split_n, const char* __restrict__ W, int M, int N,
                                                        int K, long lda_b, long ldw_b, int a_mod, tcdiff_tile_epi e) {
    typedef typename P::elem_t T;
    constexpr int ES = sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char smem[];  // NS x 32 KB (the epilogue restages through 64 KB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    // 1-D grid, XCD-remapped: an XCD owns a contiguous range of row panels with all their column tiles
    const int ntn = (N + 127) / 128;
    // split-K (TC_EPI_ATOMIC_F32): `k_splits` workgroups share a tile, each with a contiguous range of k-tiles
    const int nsplit = e.k_splits > 1 ? e.k_splits : 1;
    const int ntiles = gridDim.x / nsplit;
    const int split = blockIdx.x / ntiles;
    const int tile_id = xcd_remap(blockIdx.x - split * ntiles, ntiles);
    const int m0 = (tile_id / ntn) * 128, n0 = (tile_id % ntn) * 128;
    const char* Ause = (A2 != nullptr && n0 >= split_n) ? A2 : A;

    // acc[i][j]: operand roles swapped -> column (lane & 31) = output ROW m, rows (registers) = output COLUMN n
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

    const int nk_all = K / P::KT;
    const int kt0 = (int)(((long)nk_all * split) / nsplit);
    const int nk = (int)(((long)nk_all * (split + 1)) / nsplit) - kt0;      // >= 1: the launcher keeps splits <= k-tiles
    constexpr int STAGE = 2 * 128 * TC_ROWB;  // [A tile | W tile]
    constexpr int WOFF = 128 * TC_ROWB;

    TC_STAMP_AT(0);
    // tile t of the k-loop -> slot t % NS; past the end the last tile is re-issued (never read), so that the number of
    // DMA instructions in flight behind a tile is the same in every trip: 8 per tile and wave
    auto issue = [&](int t) {
        const int tt = kt0 + (t < nk ? t : nk - 1);
        char* dst = smem + (t % NS) * STAGE;
        if constexpr (TN) {
            // column block u (128 bytes of m / n) of the k-tile's KT rows -> LDS [u][KT][128 B]; 2 (bf16) or 4 (f32) per operand
#pragma unroll
            for (int u = 0; u < ES; ++u) {
                stage_glds<P::KT, 4>(dst + u * (P::KT * TC_ROWB), Ause + (long)m0 * ES + u * TC_ROWB, lda_b, tt * P::KT, K, 0,
                                     wave, lane);
                stage_glds<P::KT, 4>(dst + WOFF + u * (P::KT * TC_ROWB), W + (long)n0 * ES + u * TC_ROWB, ldw_b, tt * P::KT, K,
                                     0, wave, lane);
            }
        } else {
            stage_glds<128, 4>(dst, Ause + (long)tt * TC_ROWB, lda_b, m0, M, a_mod, wave, lane);
            stage_glds<128, 4>(dst + WOFF, W + (long)tt * TC_ROWB, ldw_b, n0, N, 0, wave, lane);
        }
    };
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) issue(t);
    TC_STAMP_AT(1);
    TC_STAMP_AT(2);

    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed (this wave's part: all but the 8 (NS - 2) younger DMAs; the barrier: everyone's part), and
        // every wave is out of tile kt - 1, whose slot the next DMA overwrites
        if (NS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NS == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        __syncthreads();
        issue(kt + NS - 1);
        const int cur = kt % NS;
        const char* ta = smem + cur * STAGE + (wm * 64) * TC_ROWB;       // TN: the 8-KB [KT][64 columns] tile of this wave's m range
        const char* tw = smem + cur * STAGE + WOFF + (wn * 64) * TC_ROWB;
        if constexpr (TN) {
            typedef AttnCfg<P> C;
#pragma unroll
            for (int k32 = 0; k32 < C::NKT; ++k32)
#pragma unroll
                for (int st = 0; st < C::PV_STEPS; ++st) {
                    u32x4 fa[2], fw[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) fa[i] = v_frag<P>(ta, i, k32, st, lane);
#pragma unroll
                    for (int j = 0; j < 2; ++j) fw[j] = v_frag<P>(tw, j, k32, st, lane);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) P::mma(acc[i][j], fw[j], fa[i]);
                }
            continue;
        }
        if constexpr (P::IS_X3) {     // split-bf16: k-steps in pairs, one K = 16 MFMA triple per pair (common.h MmaBF16x3::mma2)
#pragma unroll
            for (int ks = 0; ks < 4; ks += 2) {
                u32x4 fa[2][2], fw[2][2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) fa[i][q] = lds_frag(ta, i * 32 + r, 2 * (ks + q) + h);
#pragma unroll
                    for (int j = 0; j < 2; ++j) fw[j][q] = lds_frag(tw, j * 32 + r, 2 * (ks + q) + h);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) P::mma2(acc[i][j], fw[j][0], fw[j][1], fa[i][0], fa[i][1]);
            }
            continue;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 fa[2], fw[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = lds_frag(ta, i * 32 + r, 2 * ks + h);
#pragma unroll
            for (int j = 0; j < 2; ++j) fw[j] = lds_frag(tw, j * 32 + r, 2 * ks + h);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) P::mma(acc[i][j], fw[j], fa[i]);
        }
    }
    sync_dma();  // the re-issued tail DMAs have landed and every wave is out of the last tile: smem becomes the output stage

    TC_STAMP_AT(3);
    // ---- epilogue --------------------------------------------------------------------------------
    // this lane: output rows m = m0 + wm*64 + i*32 + r; columns n = n0 + wn*64 + j*32 + 8g + 4h + {0..3}
    const bool qkv = e.mode == TC_EPI_QKV_HEADS;
    float colscale = 1.0f;
    if (qkv && n0 < e.n_q) colscale = e.scale_q;  // n_q is a multiple of 128: a block is all-Q or no-Q

    if (e.mode == TC_EPI_STORE_F32) {
        float* out = reinterpret_cast<float*>(e.out);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wm * 64 + i * 32 + r;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + wn * 64 + j * 32 + 8 * g + 4 * h;
                    if (m >= M || n >= N) continue;
                    float v[4];
                    f32x4_t bq = {0.f, 0.f, 0.f, 0.f};
                    if (e.bias) {
                        if (n + 3 < N) bq = *reinterpret_cast<const f32x4_t*>(e.bias + n);
                        else
                            for (int t = 0; t < 4; ++t) if (n + t < N) bq[t] = e.bias[n + t];
                    }
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = act_ct<ACT>(acc[i][j][4 * g + t] + bq[t], e.act);
                    float* dst = out + (long)m * e.ldc + n;
                    if (n + 3 < N && (e.ldc & 3) == 0) {
                        f32x4_t pk = {v[0], v[1], v[2], v[3]};
                        *reinterpret_cast<f32x4_t*>(dst) = pk;
                    } else {
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            if (n + t < N) dst[t] = v[t];
                    }
                }
        }
        return;
    }

    if (e.mode == TC_EPI_ATOMIC_F32) {
        // out[m][n] += acc: the tile goes through LDS (two passes of 64 rows, fp32) so that every atomic wave-instruction
        // covers 64 CONSECUTIVE floats of one output row -- 256 contiguous bytes, the shape the memory-side atomic units
        // take at full rate (MI355X_MICROARCH.md, Global float atomics); a lane-per-row scatter of 16-byte pieces is ~17x slower.
        constexpr int RSF = 128 * 4 + 16;
        float* out = reinterpret_cast<float*>(e.out);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (wm == pass) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ml = i * 32 + r;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int nl = wn * 64 + j * 32 + 8 * g + 4 * h;
                            const f32x4_t pk = {acc[i][j][4 * g + 0], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                            *reinterpret_cast<f32x4_t*>(smem + ml * RSF + nl * 4) = pk;
                        }
                }
            }
            __syncthreads();
#pragma unroll 4
            for (int c = tid; c < 64 * 128; c += 256) {
                const int row = c >> 7, col = c & 127;
                const int m = m0 + pass * 64 + row, n = n0 + col;
                if (m < M && n < N)
                    unsafeAtomicAdd(out + (long)m * e.ldc + n, *reinterpret_cast<const float*>(smem + row * RSF + col * 4));
            }
            if (pass == 0) __syncthreads();
        }
        return;
    }

    // T-typed outputs: stage the tile in LDS as [rows][128 cols] T with a padded row stride, then write whole
    // 16-byte chunks of contiguous output rows.  f32 needs two passes of 64 rows to fit the 64 KB.
    constexpr int RS = 128 * ES + 16;            // staged row stride in bytes
    constexpr int NPASS = ES / 2;                // bf16: 1 pass of 128 rows; f32: 2 passes of 64 rows
    constexpr int PROWS = 128 / NPASS;
    constexpr int CPR = 128 * ES / 16;           // 16-B chunks per staged row
    constexpr int EPC = 16 / ES;                 // elements per chunk
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        if (NPASS == 1 || wm == pass) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ml = (NPASS == 1 ? wm * 64 : 0) + i * 32 + r;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nl = wn * 64 + j * 32 + 8 * g + 4 * h;
                        const int n = n0 + nl;
                        float v[4];
                        f32x4_t bq = {0.f, 0.f, 0.f, 0.f};
                        if (e.bias) {
                            if (n + 3 < N) bq = *reinterpret_cast<const f32x4_t*>(e.bias + n);
                            else
                                for (int t = 0; t < 4; ++t) if (n + t < N) bq[t] = e.bias[n + t];
                        }
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] = acc[i][j][4 * g + t] + bq[t];
                        act4_ct<ACT>(v, e.act);
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] *= colscale;
                        char* dst = smem + ml * RS + nl * ES;
                        if (P::IS_BF16) {
                            uint2 pk;
                            pk.x = pack_bf2(v[0], v[1]);
                            pk.y = pack_bf2(v[2], v[3]);
                            *reinterpret_cast<uint2*>(dst) = pk;
                        } else {
                            const f32x4_t pk = {v[0], v[1], v[2], v[3]};      // columns nl .. nl + 3: one whole chunk
                            *reinterpret_cast<u32x4*>(dst) = P::chunk_from4(pk);
                        }
                    }
            }
        }
        __syncthreads();
        TC_STAMP_AT(4);
#pragma unroll 4
        for (int c = tid; c < PROWS * CPR; c += 256) {
            const int row = c / CPR, ch = c % CPR;
            const int m = m0 + pass * PROWS + row;
            const int n = n0 + ch * EPC;
            if (m >= M || n >= N) continue;
            const u32x4 val = *reinterpret_cast<const u32x4*>(smem + row * RS + ch * 16);
            T* dst;
            if (!qkv) {
                dst = reinterpret_cast<T*>(e.out) + (long)m * e.ldc + n;
            } else {
                // column n -> (Q | K | V, head, d); row m -> (sequence, token): head-major images for attention
                const int which = n < e.n_q ? 0 : (n < e.n_q + e.n_k ? 1 : 2);
                const int nn = n - (which == 0 ? 0 : (which == 1 ? e.n_q : e.n_q + e.n_k));
                int head = nn >> 6;
                const int d = nn & 63;
                const int seq = m / e.L + e.seq_off, tok = m % e.L + e.tok_off;
                T* base = reinterpret_cast<T*>(which == 0 ? e.out : (which == 1 ? e.out_k : e.out_v));
                if (e.hgroup > 0) {          // several images side by side in the columns (the K / V of all decoder layers' cross-attention)
                    base += (long)(head / e.hgroup) * e.hgroup_stride;
                    head = head % e.hgroup;
                }
                dst = base + (((long)seq * e.H + head) * e.Lp + tok) * 64 + d;
            }
            if (!qkv && (e.out2 || e.act_src)) {
                // training step, fused activation (the launcher guarantees whole aligned chunks).  Forward (out2): `out` keeps the
                // pre-activation a (T) for the backward, out2 = T(dropout(act2(a))) -- nn.Linear + activation + nn.Dropout of
                // model/model.py:399-400,244,522-528 in one pass.  Backward (act_src): the tile is dY of the activation's output;
                // out = T(dY * mask / (1 - p) * act2'(a)) with a read from act_src.  Both on the T-rounded values, as the separate
                // tcdiff_act_drop(_bwd) kernels do; dropout index = m * N + n (train_common.h).
                const DropCtx dc = drop_ctx(e.drop_seed, e.drop_site, e.drop_thr, e.drop_scale);
                u32x4 src = val;
                if (e.act_src) src = *reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(e.act_src) + (long)m * e.ld_src + n);
                else *reinterpret_cast<u32x4*>(dst) = val;
                float x[EPC], y[EPC];
                unpack_chunk<P>(src, x);
                if (e.act_src) unpack_chunk<P>(val, y);
                // the activation is chosen by WAVE-UNIFORM branches around whole loops: a per-element switch is if-converted
                // and every lane then evaluates erf, tanh, log1p and exp for every element (common.h, act_ct)
                float f[EPC];
                const float sc = e.drop_thr ? e.drop_scale : 1.0f;
                if (e.act_src) {
#pragma unroll
                    for (int t = 0; t < EPC; ++t) y[t] *= sc;          // (dY * scale) * act'(a): the rounding order of tcdiff_act_drop_bwd
                    if (e.act2 == ACT_GELU) {
#pragma unroll
                        for (int t = 0; t < EPC; ++t) f[t] = y[t] * gelu_grad(x[t]);
                    } else if (e.act2 == ACT_RELU) {
#pragma unroll
                        for (int t = 0; t < EPC; ++t) f[t] = x[t] > 0.0f ? y[t] : 0.0f;
                    } else {
#pragma unroll
                        for (int t = 0; t < EPC; ++t) f[t] = y[t] * act_grad(x[t], e.act2);
                    }
                } else {
                    if (e.act2 == ACT_GELU) {
#pragma unroll
                        for (int t = 0; t < EPC; ++t) f[t] = gelu_erf(x[t]);
                    } else if (e.act2 == ACT_RELU) {
#pragma unroll
                        for (int t = 0; t < EPC; ++t) f[t] = fmaxf(x[t], 0.0f);
                    } else {
#pragma unroll
                        for (int t = 0; t < EPC; ++t) f[t] = apply_act(x[t], e.act2);
                    }
                }
                if (!e.act_src) {
#pragma unroll
                    for (int t = 0; t < EPC; ++t) f[t] *= sc;
                }
                if (e.drop_thr) {
#pragma unroll
                    for (int t = 0; t < EPC; ++t)
                        f[t] = drop_keep(dc, (uint32_t)m * (uint32_t)N + (uint32_t)(n + t)) ? f[t] : 0.0f;
                }
#pragma unroll
                for (int t = 0; t < EPC; ++t) y[t] = f[t];
                T* d2 = e.act_src ? dst : reinterpret_cast<T*>(e.out2) + (long)m * e.ldc2 + n;
                *reinterpret_cast<u32x4*>(d2) = pack_chunk<P>(y);
            } else if (n + EPC <= N && (qkv || ((long)e.ldc * ES) % 16 == 0)) {
                *reinterpret_cast<u32x4*>(dst) = val;
            } else {
                const T* sv = reinterpret_cast<const T*>(smem + row * RS + ch * 16);
                for (int t = 0; t < EPC; ++t)
                    if (n + t < N) dst[t] = sv[t];
            }
        }
        if (pass + 1 < NPASS) __syncthreads();
    }
    TC_STAMP_AT(5);
}

// =================================================================================================
// gemm_tn_grouped: the weight gradients of SEVERAL nn.Linears in one launch, work split evenly over the CUs
// =================================================================================================
// A weight gradient is a small output (512 x 512 ... 1024 x 1536) over a long contraction (14 400 token rows): one launch
// per linear needs ~16 workgroups per tile to fill the chip, and then each of them runs 14 k-tiles and adds a 64-KB tile
// with atomics -- the launch is two thirds prologue + atomics (512 x 512: 30 us for 9 us of k-loop; the memory-side adders
// take ~1 MB per us).  The backward of a decoder layer produces seven such gradients whose operands all exist by the end
// of the layer, so they are done together: the (tile, k-tile) work units of all problems form one list, cut into equal
// contiguous shares, one per workgroup (one workgroup per CU).  A share crosses at most a few tile boundaries; per tile
// segment the workgroup runs the TN k-loop of gemm_tile_kernel over its k-range and adds the tile once.  Atomic volume drops
// from (16 splits x tiles) to (~1.5 x tiles), every CU has the same number of k-tiles.
template <class P>
__global__ __launch_bounds__(256) void gemm_tn_grouped_kernel(tcdiff_tn_group g) {
    typedef typename P::elem_t T;
    typedef AttnCfg<P> C;
    constexpr int ES = sizeof(T), NS = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    constexpr int STAGE = 2 * 128 * TC_ROWB, WOFF = 128 * TC_ROWB;
    // Unit order = (problem, k-chunk of g.kc k-tiles, tile, k-tile inside the chunk), shares handed out so that an XCD owns a
    // contiguous range of them: at any moment the 32 CUs of an XCD work on neighbouring tiles of the SAME chunk of token rows,
    // i.e. on the same few column blocks of dY and X, which then come out of that XCD's L2.  (Tile-major order over the whole
    // contraction -- neighbours at unrelated token rows -- ran at HBM speed: 196 us for a decoder layer's group, 6.6 TB/s.)
    int u = xcd_remap(blockIdx.x, gridDim.x) * g.units_per_wg;
    const int u_end = min(u + g.units_per_wg, g.total_units);
    while (u < u_end) {
        int pi = 0;
        while (pi + 1 < g.n_prob && g.p[pi + 1].unit0 <= u) ++pi;            // wave-uniform scan over <= 16 problems
        const tcdiff_tn_problem pr = g.p[pi];
        const int tiles_n = pr.N / 128, tiles = (pr.M / 128) * tiles_n;
        const int lu = u - pr.unit0;
        int ch = lu / (tiles * g.kc);
        const int nch = (pr.nk + g.kc - 1) / g.kc;
        ch = ch < nch ? ch : nch - 1;
        const int clen = min(g.kc, pr.nk - ch * g.kc);                       // k-tiles in this chunk
        const int local = lu - ch * tiles * g.kc;
        const int tile = local / clen, kin = local - tile * clen;
        const int kt0 = ch * g.kc + kin;
        const int nk = min(clen - kin, u_end - u);
        const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
        const char* A = reinterpret_cast<const char*>(pr.A);
        const char* B = reinterpret_cast<const char*>(pr.B);
        const long lda_b = (long)pr.lda * ES, ldb_b = (long)pr.ldb * ES;
        const int Kt = pr.nk * P::KT;
        f32x16_t acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;
        auto issue = [&](int t) {
            const int tt = kt0 + (t < nk ? t : nk - 1);
            char* dst = smem + (t % NS) * STAGE;
#pragma unroll
            for (int c = 0; c < ES; ++c) {
                stage_glds<P::KT, 4>(dst + c * (P::KT * TC_ROWB), A + (long)m0 * ES + c * TC_ROWB, lda_b, tt * P::KT, Kt, 0, wave, lane);
                stage_glds<P::KT, 4>(dst + WOFF + c * (P::KT * TC_ROWB), B + (long)n0 * ES + c * TC_ROWB, ldb_b, tt * P::KT, Kt, 0,
                                     wave, lane);
            }
        };
#pragma unroll
        for (int t = 0; t < NS - 1; ++t) issue(t);
        for (int kt = 0; kt < nk; ++kt) {
            asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // tile kt landed: all but the 8 (NS - 2) younger DMAs of this wave
            __syncthreads();
            issue(kt + NS - 1);
            const int cur = kt % NS;
            const char* ta = smem + cur * STAGE + (wm * 64) * TC_ROWB;
            const char* tw = smem + cur * STAGE + WOFF + (wn * 64) * TC_ROWB;
#pragma unroll
            for (int k32 = 0; k32 < C::NKT; ++k32)
#pragma unroll
                for (int st = 0; st < C::PV_STEPS; ++st) {
                    u32x4 fa[2], fw[2];
#pragma unroll
                    for (int i = 0; i < 2; ++i) fa[i] = v_frag<P>(ta, i, k32, st, lane);
#pragma unroll
                    for (int j = 0; j < 2; ++j) fw[j] = v_frag<P>(tw, j, k32, st, lane);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) P::mma(acc[i][j], fw[j], fa[i]);
                }
        }
        sync_dma();          // the re-issued tail DMAs have landed, every wave is out of the last tile: smem becomes the output stage
        // out[m][n] += acc through LDS: 64 consecutive floats per atomic wave-instruction (gemm_tile_kernel, TC_EPI_ATOMIC_F32)
        constexpr int RSF = 128 * 4 + 16;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            if (wm == pass) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int ml = i * 32 + r;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4) {
                            const int nl = wn * 64 + j * 32 + 8 * q4 + 4 * h;
                            const f32x4_t pk = {acc[i][j][4 * q4 + 0], acc[i][j][4 * q4 + 1], acc[i][j][4 * q4 + 2], acc[i][j][4 * q4 + 3]};
                            *reinterpret_cast<f32x4_t*>(smem + ml * RSF + nl * 4) = pk;
                        }
                }
            }
            __syncthreads();
#pragma unroll 4
            for (int c = tid; c < 64 * 128; c += 256) {
                const int row = c >> 7, col = c & 127;
                unsafeAtomicAdd(pr.out + (long)(m0 + pass * 64 + row) * pr.ldc + n0 + col,
                                *reinterpret_cast<const float*>(smem + row * RSF + col * 4));
            }
            __syncthreads();          // the stage is reused: by the second pass, then by the next segment's DMA
        }
        u += nk;
    }
}

// =================================================================================================
// gemm_rowln: 64 x 512 tiles, 512 threads; row-complete epilogue
// =================================================================================================
#define ROWLN_SMEM (2 * (64 + 512) * TC_ROWB)

template <class P>
__global__ __launch_bounds__(512) void gemm_rowln_kernel(const char* __restrict__ A, const char* __restrict__ W,
                                                         int M, int K, long lda_b, long ldw_b, int a_mod,
                                                         tcdiff_row_epi e) {
    typedef typename P::elem_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    // grid = row tiles x groups; an XCD owns a contiguous range of rows (as gemm_tile); group g (tcdiff_hip.h) shifts
    // the weight rows, the bias and the output row offset.  Measured and dropped: shorter tiles for launches with few
    // rows (48 / 32 valid rows, to cover the chip: 2.38 / 2.46 ms per step against 2.25) and, in gemm_tile, 8-wave
    // workgroups that compute two tiles so that small launches pack two tiles per CU (2.33 against 2.24).
    const int ntile = (M + 63) / 64;
    const int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int grp = blk / ntile;
    const int m0 = (blk - grp * ntile) * 64;
    W += (long)grp * 512 * ldw_b;
    if (e.bias) e.bias += grp * 512;
    e.out_add += grp;

    constexpr int STAGE = (64 + 512) * TC_ROWB;  // [A tile | W tile]
    constexpr int WOFF = 64 * TC_ROWB;

    // Residual rows of this wave's 8 output rows (fp32, 2 KB each) are fetched into registers BEFORE the main loop:
    // the per-CU fabric rate (~25 GB/s) makes these reads as expensive as the GEMM itself, and the main loop only
    // uses the L2 -> LDS DMA path, so they overlap completely.
    const int f = e.flags;
    const int c0 = 4 * lane, c1 = 256 + 4 * lane;
    f32x4_t xra[8], xrb[8];
    if (f & (TC_ROW_FILM | TC_ROW_RES)) {
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            int m = m0 + wave * 8 + rr;
            m = m < M ? m : M - 1;
            const int mr = e.xres_mod > 0 ? m % e.xres_mod : m;
            xra[rr] = *reinterpret_cast<const f32x4_t*>(e.xres + (long)mr * 512 + c0);
            xrb[rr] = *reinterpret_cast<const f32x4_t*>(e.xres + (long)mr * 512 + c1);
        }
    }

    f32x16_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.0f;

    const int nk = K / P::KT;
    TC_STAMP_AT(0);
    stage_glds<64, 8>(smem, A, lda_b, m0, M, a_mod, wave, lane);
    stage_glds<512, 8>(smem + WOFF, W, ldw_b, 0, 512, 0, wave, lane);
    sync_dma();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            char* nxt = smem + (cur ^ 1) * STAGE;
            stage_glds<64, 8>(nxt, A + (long)(kt + 1) * TC_ROWB, lda_b, m0, M, a_mod, wave, lane);
            stage_glds<512, 8>(nxt + WOFF, W + (long)(kt + 1) * TC_ROWB, ldw_b, 0, 512, 0, wave, lane);
        }
        const char* ta = smem + cur * STAGE + (wm * 32) * TC_ROWB;
        const char* tw = smem + cur * STAGE + WOFF + (wn * 128) * TC_ROWB;
        if constexpr (P::IS_X3) {     // split-bf16: k-steps in pairs (see gemm_tile_kernel)
#pragma unroll
            for (int ks = 0; ks < 4; ks += 2) {
                const u32x4 fa0 = lds_frag(ta, r, 2 * ks + h), fa1 = lds_frag(ta, r, 2 * ks + 2 + h);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const u32x4 fw0 = lds_frag(tw, j * 32 + r, 2 * ks + h), fw1 = lds_frag(tw, j * 32 + r, 2 * ks + 2 + h);
                    P::mma2(acc[j], fa0, fa1, fw0, fw1);
                }
            }
        } else {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 fa = lds_frag(ta, r, 2 * ks + h);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x4 fw = lds_frag(tw, j * 32 + r, 2 * ks + h);
                P::mma(acc[j], fa, fw);
            }
        }
        }
        sync_dma();
    }

    TC_STAMP_AT(1);
    // ---- phase 1: dump the fp32 accumulator tile to LDS as [64 rows][512 cols] --------------------------
    // lane: column n = wn*128 + j*32 + r, rows wm*32 + acc_row(q,h).  32 lanes write 32 consecutive floats.
    float* tile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = wn * 128 + j * 32 + r;
#pragma unroll
        for (int q = 0; q < 16; ++q) tile[(wm * 32 + acc_row(q, h)) * 512 + n] = acc[j][q];
    }
    __syncthreads();

    TC_STAMP_AT(2);
    // ---- phase 2: one wave per row; lane owns columns [4l,4l+4) and [256+4l, 256+4l+4) -------------------
    f32x4_t bias0 = {0, 0, 0, 0}, bias1 = {0, 0, 0, 0};
    if (f & TC_ROW_BIAS) {
        bias0 = *reinterpret_cast<const f32x4_t*>(e.bias + c0);
        bias1 = *reinterpret_cast<const f32x4_t*>(e.bias + c1);
    }
    f32x4_t g1a = {0, 0, 0, 0}, g1b = g1a, b1a = g1a, b1b = g1a, g2a = g1a, g2b = g1a, b2a = g1a, b2b = g1a;
    if (f & TC_ROW_LN_POST) {
        g1a = *reinterpret_cast<const f32x4_t*>(e.ln_g + c0);
        g1b = *reinterpret_cast<const f32x4_t*>(e.ln_g + c1);
        b1a = *reinterpret_cast<const f32x4_t*>(e.ln_b + c0);
        b1b = *reinterpret_cast<const f32x4_t*>(e.ln_b + c1);
    }
    if (f & TC_ROW_NEXT_LN) {
        g2a = *reinterpret_cast<const f32x4_t*>(e.nln_g + c0);
        g2b = *reinterpret_cast<const f32x4_t*>(e.nln_g + c1);
        b2a = *reinterpret_cast<const f32x4_t*>(e.nln_b + c0);
        b2b = *reinterpret_cast<const f32x4_t*>(e.nln_b + c1);
    }

    auto normalize = [&](f32x4_t& va, f32x4_t& vb, float eps, const f32x4_t& ga, const f32x4_t& gb,
                         const f32x4_t& ba, const f32x4_t& bb) {
        const float mean = wave_sum((va[0] + va[1]) + (va[2] + va[3]) + (vb[0] + vb[1]) + (vb[2] + vb[3])) * (1.0f / 512.0f);
        float ss = 0.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float da = va[t] - mean, db = vb[t] - mean;
            ss += da * da + db * db;
        }
        const float rstd = rsqrtf(wave_sum(ss) * (1.0f / 512.0f) + eps);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            va[t] = (va[t] - mean) * rstd * ga[t] + ba[t];
            vb[t] = (vb[t] - mean) * rstd * gb[t] + bb[t];
        }
    };
    auto store_T = [&](void* base, long row, const f32x4_t& va, const f32x4_t& vb) {
        T* p = reinterpret_cast<T*>(base) + row * 512;
        if (P::IS_BF16) {
            uint2 pa, pb;
            pa.x = pack_bf2(va[0], va[1]); pa.y = pack_bf2(va[2], va[3]);
            pb.x = pack_bf2(vb[0], vb[1]); pb.y = pack_bf2(vb[2], vb[3]);
            *reinterpret_cast<uint2*>(p + c0) = pa;
            *reinterpret_cast<uint2*>(p + c1) = pb;
        } else {
            *reinterpret_cast<u32x4*>(p + c0) = P::chunk_from4(va);
            *reinterpret_cast<u32x4*>(p + c1) = P::chunk_from4(vb);
        }
    };

    // Global operands of a row (FiLM scale/shift, residual, rotary cos/sin) do not depend on the arithmetic, so the
    // loads of row rr+1 are issued before row rr is reduced: their latency hides under the shuffle chains.
    struct RowIn { f32x4_t sa, sb, ha, hb, ca, cb; };
    auto fetch = [&](int rr, RowIn& in) {
        const int m = m0 + wave * 8 + rr;
        if (rr >= 8 || m >= M) return;
        if (f & TC_ROW_FILM) {
            const float* fp = e.film + (long)(m / e.L) * e.film_ld;
            in.sa = *reinterpret_cast<const f32x4_t*>(fp + c0); in.sb = *reinterpret_cast<const f32x4_t*>(fp + c1);
            in.ha = *reinterpret_cast<const f32x4_t*>(fp + 512 + c0); in.hb = *reinterpret_cast<const f32x4_t*>(fp + 512 + c1);
        }
        if (f & TC_ROW_STORE_ROT) {
            const int pos = (int)(((long)m * e.out_mul + e.out_add) % e.L);
            in.ca = *reinterpret_cast<const f32x4_t*>(e.rope + (long)pos * 512 + c0);  // cos0 sin0 cos1 sin1
            in.cb = *reinterpret_cast<const f32x4_t*>(e.rope + (long)pos * 512 + c1);
        }
    };
    auto process = [&](int rr, const RowIn& in, const f32x4_t& xa, const f32x4_t& xb) {
        const int row = wave * 8 + rr;
        const int m = m0 + row;
        if (m >= M) return;
        f32x4_t va = *reinterpret_cast<const f32x4_t*>(tile + row * 512 + c0);
        f32x4_t vb = *reinterpret_cast<const f32x4_t*>(tile + row * 512 + c1);
        va += bias0;
        vb += bias1;
        if (f & TC_ROW_LN_POST) normalize(va, vb, e.ln_eps, g1a, g1b, b1a, b1b);
        if (f & TC_ROW_FILM) {
            va = (in.sa + 1.0f) * va + in.ha;
            vb = (in.sb + 1.0f) * vb + in.hb;
        }
        if (f & (TC_ROW_FILM | TC_ROW_RES)) {
            va = xa + va;
            vb = xb + vb;
        }
        const long mo = (long)m * e.out_mul + e.out_add;
        if (f & TC_ROW_STORE_X) {
            *reinterpret_cast<f32x4_t*>(e.xout + mo * 512 + c0) = va;
            *reinterpret_cast<f32x4_t*>(e.xout + mo * 512 + c1) = vb;
        }
        if (!(f & TC_ROW_NEXT_LN)) {
            if (f & TC_ROW_STORE_H) store_T(e.hout, mo, va, vb);
            return;
        }
        normalize(va, vb, e.nln_eps, g2a, g2b, b2a, b2b);
        if (f & TC_ROW_STORE_H) store_T(e.hout, mo, va, vb);
        if (f & TC_ROW_STORE_ROT) {
            f32x4_t ya, yb;
            ya[0] = va[0] * in.ca[0] - va[1] * in.ca[1]; ya[1] = va[1] * in.ca[0] + va[0] * in.ca[1];
            ya[2] = va[2] * in.ca[2] - va[3] * in.ca[3]; ya[3] = va[3] * in.ca[2] + va[2] * in.ca[3];
            yb[0] = vb[0] * in.cb[0] - vb[1] * in.cb[1]; yb[1] = vb[1] * in.cb[0] + vb[0] * in.cb[1];
            yb[2] = vb[2] * in.cb[2] - vb[3] * in.cb[3]; yb[3] = vb[3] * in.cb[2] + vb[2] * in.cb[3];
            store_T(e.rout, mo, ya, yb);
        }
    };
    RowIn inA, inB;
    fetch(0, inA);
#pragma unroll
    for (int rr = 0; rr < 8; rr += 2) {
        fetch(rr + 1, inB);
        process(rr, inA, xra[rr], xrb[rr]);
        fetch(rr + 2, inA);
        process(rr + 1, inB, xra[rr + 1], xrb[rr + 1]);
    }
    TC_STAMP_AT(3);
}

// =================================================================================================
// C ABI
// =================================================================================================
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- small-M products: K dealt to the waves of a workgroup ---------------------------------------------------------------------
// The sampler's input / fusion projections on a small job (model/model.py:560-561: M = frames of a few clips, 150 for one 3-dancer
// clip; N = 1024, K = 576 / 1024) are 16 tiles of gemm_tile_kernel: 16 of 256 CUs, each walking 16 k-tiles between barriers --
// 17 us for 0.3 GFLOP (profiles/r06_kernel_stats_small_job_ddim50_1clip.csv).  Here one workgroup owns a 32 x 32 output tile and
// its 8 waves SPLIT K (k-step ks on wave ks % 8): a wave loads its operand fragments global -> registers in the MFMA layout (lane
// (c, g): row c, 16 bytes at k = 32 ks + 8 g; no LDS staging, no barrier inside the k loop), all of them in flight at once for
// K <= 1024, and the eight partial tiles meet once in LDS.  160 workgroups for 150 x 1024; the launch is one memory round trip,
// 4 k-steps of MFMA and one exchange.  Transposed product (weights on the MFMA's row operand): a lane ends up with four consecutive
// columns of one output row.  bf16 operands, TC_EPI_STORE_T / TC_EPI_STORE_F32 epilogues without the training extras.
template <bool F32OUT>
__global__ __launch_bounds__(512) void gemm_small_kernel(const char* A, const char* W, int M, int N, int K, long lda_b, long ldw_b,
                                                         int a_mod, tcdiff_tile_epi e) {
    __shared__ f32x4_t red[8][4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nct = N >> 5;
    const int ct = blockIdx.x % nct, rt = blockIdx.x / nct;       // the column tiles of a row tile are neighbours: its A rows stay in L2
    const int c = lane & 15, g = lane >> 4;
    const char *ap[2], *wp[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int m = rt * 32 + 16 * i + c;
        m = m < M ? m : M - 1;
        if (a_mod > 0) m %= a_mod;
        ap[i] = A + (long)m * lda_b + g * 16;
        wp[i] = W + (long)(ct * 32 + 16 * i + c) * ldw_b + g * 16;
    }
    f32x4_t acc[2][2];                                            // [n tile][m tile]
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i >> 1][i & 1] = f32x4_t{0, 0, 0, 0};
    const int nk = K >> 5;
#pragma unroll 4
    for (int ks = wave; ks < nk; ks += 8) {
        const u32x4 a0 = *reinterpret_cast<const u32x4*>(ap[0] + ks * 64), a1 = *reinterpret_cast<const u32x4*>(ap[1] + ks * 64);
        const u32x4 w0 = *reinterpret_cast<const u32x4*>(wp[0] + ks * 64), w1 = *reinterpret_cast<const u32x4*>(wp[1] + ks * 64);
        acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w0), __builtin_bit_cast(bf16x8_t, a0), acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w0), __builtin_bit_cast(bf16x8_t, a1), acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w1), __builtin_bit_cast(bf16x8_t, a0), acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w1), __builtin_bit_cast(bf16x8_t, a1), acc[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][i][lane] = acc[i >> 1][i & 1];
    __syncthreads();
    if (wave >= 4) return;
    f32x4_t s = red[0][wave][lane];                               // wave w (< 4) finishes tile w: fixed summation order
#pragma unroll
    for (int w = 1; w < 8; ++w) s += red[w][wave][lane];
    const int m = rt * 32 + 16 * (wave & 1) + c, n = ct * 32 + 16 * (wave >> 1) + 4 * g;
    if (m >= M) return;
    float v[4] = {s[0], s[1], s[2], s[3]};
    if (e.bias) {
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(e.bias + n);
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] += b[t];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = apply_act(v[t], e.act);
    if (F32OUT) {
        *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(e.out) + (long)m * e.ldc + n) = f32x4_t{v[0], v[1], v[2], v[3]};
    } else {
        uint2 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(e.out) + (long)m * e.ldc + n) = pk;
    }
}

// whether launch_tile hands a product to gemm_small_kernel: asked for (tcdiff_tile_epi.small_m), bf16, plain epilogue, and a 128 x 128 tiling that would leave at least
// three quarters of the CUs idle while the 32 x 32 tiling still fits the machine once or twice (TCDIFF_GEMM_SMALL=0: never)
static bool small_product(int dtype, const void* A2, int M, int N, int K, const tcdiff_tile_epi& e, int n_cu) {
    static const bool off = [] { const char* v = getenv("TCDIFF_GEMM_SMALL"); return v && v[0] == '0'; }();
    if (off || !e.small_m || dtype != TC_DTYPE_BF16 || A2 || (e.mode != TC_EPI_STORE_T && e.mode != TC_EPI_STORE_F32) || e.out2 || e.act_src)
        return false;
    if (N % 32 || K % 32 || K > 2048 || e.ldc % 4 || (e.bias && !aligned16(e.bias))) return false;
    const long big = (long)((N + 127) / 128) * ((M + 127) / 128), small = (long)(N / 32) * ((M + 31) / 32);
    return big * 4 <= n_cu && small <= 2L * n_cu;
}

static int launch_tile(int dtype, const void* A, const void* A2, int split_n, const void* W, int M, int N, int K,
                       int lda, int ldw, int a_mod, const tcdiff_tile_epi* epi, hipStream_t stream) {
    if (!A || !W || !epi || M <= 0 || N <= 0 || K <= 0) return TC_ERR_ARG;
    if (dtype != TC_DTYPE_BF16 && dtype != TC_DTYPE_F32 && dtype != TC_DTYPE_BF16X3) return TC_ERR_ARG;
    // TC_DTYPE_BF16X3: fp32 storage (every size and alignment rule of TC_DTYPE_F32), products as split-bf16 triples; the plain
    // forward epilogues only (the sampler's path: no split-K, no fused training epilogues)
    if (dtype == TC_DTYPE_BF16X3 && (epi->mode == TC_EPI_ATOMIC_F32 || epi->out2 || epi->act_src)) return TC_ERR_UNSUPPORTED;
    const int es = dtype == TC_DTYPE_BF16 ? 2 : 4;
    const int kt = dtype == TC_DTYPE_BF16 ? 64 : 32;
    if (K % kt != 0) return TC_ERR_ARG;
    if (!aligned16(A) || !aligned16(W) || (A2 && !aligned16(A2)) || ((long)lda * es) % 16 || ((long)ldw * es) % 16)
        return TC_ERR_ALIGN;
    if (A2 && (split_n % 128 != 0)) return TC_ERR_ARG;
    // staging addresses are 32-bit offsets from the operand base
    if ((long)(a_mod > 0 ? a_mod : M) * lda * es >= (1L << 32) || (long)N * ldw * es >= (1L << 32)) return TC_ERR_ARG;
    if (epi->bias && !aligned16(epi->bias)) return TC_ERR_ALIGN;
    if (epi->act < TC_ACT_NONE || epi->act > TC_ACT_SILU) return TC_ERR_ARG;
    if (epi->mode == TC_EPI_QKV_HEADS) {
        if (epi->L <= 0 || epi->Lp <= 0 || epi->H <= 0 || epi->n_q % 128 || epi->n_k % 128 || N % 64) return TC_ERR_ARG;
        if ((epi->n_q > 0 && (!epi->out || !aligned16(epi->out))) ||
            (epi->n_k > 0 && (!epi->out_k || !aligned16(epi->out_k))) ||
            (N > epi->n_q + epi->n_k && (!epi->out_v || !aligned16(epi->out_v))))
            return TC_ERR_ALIGN;
    } else {
        if (!epi->out) return TC_ERR_ARG;
        if (!aligned16(epi->out)) return TC_ERR_ALIGN;
    }
    if (epi->out2 || epi->act_src) {            // fused activation epilogues (training step): whole aligned chunks only
        const int epc = 16 / es;
        if (epi->mode != TC_EPI_STORE_T || (epi->out2 && epi->act_src) || epi->act != TC_ACT_NONE) return TC_ERR_ARG;
        if (epi->act2 < TC_ACT_NONE || epi->act2 > TC_ACT_SILU || N % epc || epi->ldc % epc) return TC_ERR_ARG;
        if (epi->out2 && (epi->ldc2 < N || epi->ldc2 % epc || !aligned16(epi->out2))) return TC_ERR_ARG;
        if (epi->act_src && (epi->ld_src < N || epi->ld_src % epc || !aligned16(epi->act_src))) return TC_ERR_ARG;
    }
    tcdiff_tile_epi e = *epi;
    if (e.mode == TC_EPI_ATOMIC_F32) {
        if (e.bias || e.act != TC_ACT_NONE || A2) return TC_ERR_ARG;
        if (e.k_splits < 1) e.k_splits = 1;
        if (e.k_splits > K / kt) e.k_splits = K / kt;
    } else {
        e.k_splits = 1;
    }
    const int actk = e.act <= TC_ACT_GELU ? e.act : 3;   // 3 = runtime choice between the setup-only Mish / SiLU
    dim3 grid(((N + 127) / 128) * ((M + 127) / 128) * e.k_splits);
    static tc_dev_state dev_state;
    const int n_cu = tc_device_once(dev_state, [](int) {
        hipError_t err = hipSuccess;
#define TC_TILE_ATTR(POL, ACTK)                                                                                    \
        if (err == hipSuccess)                                                                                     \
            err = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tile_kernel<POL, ACTK, 4>),              \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 128 * TC_ROWB);
        TC_TILE_ATTR(MmaBF16, 0) TC_TILE_ATTR(MmaBF16, 1) TC_TILE_ATTR(MmaBF16, 2) TC_TILE_ATTR(MmaBF16, 3)
        TC_TILE_ATTR(MmaF32, 0) TC_TILE_ATTR(MmaF32, 1) TC_TILE_ATTR(MmaF32, 2) TC_TILE_ATTR(MmaF32, 3)
        TC_TILE_ATTR(MmaBF16x3, 0) TC_TILE_ATTR(MmaBF16x3, 1) TC_TILE_ATTR(MmaBF16x3, 2) TC_TILE_ATTR(MmaBF16x3, 3)
#undef TC_TILE_ATTR
        return err;
    });
    if (n_cu < 0) return n_cu;
    if (small_product(dtype, A2, M, N, K, e, n_cu)) {
        const dim3 sgrid((unsigned)((N / 32) * ((M + 31) / 32)));
        if (e.mode == TC_EPI_STORE_F32)
            hipLaunchKernelGGL(gemm_small_kernel<true>, sgrid, dim3(512), 0, stream, (const char*)A, (const char*)W, M, N, K,
                               (long)lda * es, (long)ldw * es, a_mod, e);
        else
            hipLaunchKernelGGL(gemm_small_kernel<false>, sgrid, dim3(512), 0, stream, (const char*)A, (const char*)W, M, N, K,
                               (long)lda * es, (long)ldw * es, a_mod, e);
        TC_CHECK_LAUNCH();
        return TC_OK;
    }
    // about one tile per CU and a k-loop long enough to matter: four stages, one workgroup per CU
    const bool deep = (long)grid.x * 4 <= (long)n_cu * 5 && K / kt / e.k_splits >= 3;
#define TC_LAUNCH_TILE(POL, ACTK)                                                                                       \
    if (deep)                                                                                                           \
        hipLaunchKernelGGL((gemm_tile_kernel<POL, ACTK, 4>), grid, dim3(256), 4 * 2 * 128 * TC_ROWB, stream,            \
                           (const char*)A, (const char*)A2, split_n, (const char*)W, M, N, K, (long)lda * es,           \
                           (long)ldw * es, a_mod, e);                                                                    \
    else                                                                                                                \
        hipLaunchKernelGGL((gemm_tile_kernel<POL, ACTK, 2>), grid, dim3(256), 2 * 2 * 128 * TC_ROWB, stream,            \
                           (const char*)A, (const char*)A2, split_n, (const char*)W, M, N, K, (long)lda * es,           \
                           (long)ldw * es, a_mod, e)
#define TC_DISPATCH_TILE(POL)                                      \
    switch (actk) {                                                \
        case 0: TC_LAUNCH_TILE(POL, 0); break;                     \
        case 1: TC_LAUNCH_TILE(POL, 1); break;                     \
        case 2: TC_LAUNCH_TILE(POL, 2); break;                     \
        default: TC_LAUNCH_TILE(POL, 3); break;                    \
    }
    if (dtype == TC_DTYPE_BF16) {
        TC_DISPATCH_TILE(MmaBF16)
    } else if (dtype == TC_DTYPE_BF16X3) {
        TC_DISPATCH_TILE(MmaBF16x3)
    } else {
        TC_DISPATCH_TILE(MmaF32)
    }
    TC_CHECK_LAUNCH();
    return TC_OK;
}

extern "C" int tcdiff_gemm_tile(int dtype, const void* A, const void* A2, int split_n, const void* W, int M, int N,
                                int K, int lda, int ldw, int a_mod, const tcdiff_tile_epi* epi,
                                hipStream_t stream) {
    if (epi && epi->mode == TC_EPI_ATOMIC_F32) return TC_ERR_ARG;      // reached through tcdiff_gemm_splitk only
    return launch_tile(dtype, A, A2, split_n, W, M, N, K, lda, ldw, a_mod, epi, stream);
}

// dW += dY^T X and friends: out[m][n] += sum_k A[m][k] W[n][k], `splits` workgroups per tile (include/tcdiff_hip.h)
extern "C" int tcdiff_gemm_splitk(int dtype, const void* A, const void* W, int M, int N, int K, int lda, int ldw,
                                  float* out, int ldc, int splits, hipStream_t stream) {
    if (!out || ldc < N || splits < 1) return TC_ERR_ARG;
    tcdiff_tile_epi e = {};
    e.mode = TC_EPI_ATOMIC_F32;
    e.out = out;
    e.ldc = ldc;
    e.k_splits = splits;
    return launch_tile(dtype, A, nullptr, 0, W, M, N, K, lda, ldw, 0, &e, stream);
}

// out[m][n] += sum_k A[k][m] B[k][n]: the same accumulation with both operands row-major over k (include/tcdiff_hip.h)
extern "C" int tcdiff_gemm_tn(int dtype, const void* A, const void* B, int M, int N, int K, int lda, int ldb, float* out,
                              int ldc, int splits, hipStream_t stream) {
    if (!A || !B || !out || M <= 0 || N <= 0 || K <= 0 || ldc < N || splits < 1 || lda < M || ldb < N) return TC_ERR_ARG;
    if (dtype != TC_DTYPE_BF16 && dtype != TC_DTYPE_F32) return TC_ERR_ARG;
    const int es = dtype == TC_DTYPE_BF16 ? 2 : 4;
    const int kt = dtype == TC_DTYPE_BF16 ? 64 : 32;
    if (M % 128 || N % 128 || K % kt) return TC_ERR_UNSUPPORTED;       // no column clamp, no k tail: the caller repacks instead
    if (!aligned16(A) || !aligned16(B) || !aligned16(out) || ((long)lda * es) % 16 || ((long)ldb * es) % 16) return TC_ERR_ALIGN;
    if ((long)K * lda * es >= (1L << 32) || (long)K * ldb * es >= (1L << 32)) return TC_ERR_ARG;   // 32-bit staging offsets
    tcdiff_tile_epi e = {};
    e.mode = TC_EPI_ATOMIC_F32;
    e.out = out;
    e.ldc = ldc;
    e.k_splits = splits < K / kt ? splits : K / kt;
    static tc_dev_state dev_state;
    const int n_cu = tc_device_once(dev_state, [](int) {
        hipError_t a = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tile_kernel<MmaBF16, 0, 4, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 128 * TC_ROWB);
        hipError_t b = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tile_kernel<MmaF32, 0, 4, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 128 * TC_ROWB);
        return a != hipSuccess ? a : b;
    });
    if (n_cu < 0) return n_cu;
    dim3 grid((N / 128) * (M / 128) * e.k_splits);
    const bool deep = (long)grid.x * 4 <= (long)n_cu * 5 && K / kt / e.k_splits >= 3;
#define TC_LAUNCH_TN(POL, NSV)                                                                                         \
    hipLaunchKernelGGL((gemm_tile_kernel<POL, 0, NSV, true>), grid, dim3(256), NSV * 2 * 128 * TC_ROWB, stream,          \
                       (const char*)A, (const char*)nullptr, 0, (const char*)B, M, N, K, (long)lda * es, (long)ldb * es, 0, e)
    if (dtype == TC_DTYPE_BF16) {
        if (deep) TC_LAUNCH_TN(MmaBF16, 4); else TC_LAUNCH_TN(MmaBF16, 2);
    } else {
        if (deep) TC_LAUNCH_TN(MmaF32, 4); else TC_LAUNCH_TN(MmaF32, 2);
    }
#undef TC_LAUNCH_TN
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// several weight gradients in one evenly split launch (include/tcdiff_hip.h)
extern "C" int tcdiff_gemm_tn_grouped(int dtype, const tcdiff_tn_problem* probs, int n_prob, hipStream_t stream) {
    if (!probs || n_prob <= 0 || n_prob > TC_TN_MAX_PROB) return TC_ERR_ARG;
    if (dtype != TC_DTYPE_BF16 && dtype != TC_DTYPE_F32) return TC_ERR_ARG;
    const int es = dtype == TC_DTYPE_BF16 ? 2 : 4;
    const int kt = dtype == TC_DTYPE_BF16 ? 64 : 32;
    tcdiff_tn_group g = {};
    g.n_prob = n_prob;
    long units = 0;
    for (int i = 0; i < n_prob; ++i) {
        tcdiff_tn_problem p = probs[i];
        if (!p.A || !p.B || !p.out || p.M <= 0 || p.N <= 0 || p.K <= 0 || p.ldc < p.N || p.lda < p.M || p.ldb < p.N) return TC_ERR_ARG;
        if (p.M % 128 || p.N % 128 || p.K % kt) return TC_ERR_UNSUPPORTED;
        if (!aligned16(p.A) || !aligned16(p.B) || !aligned16(p.out) || ((long)p.lda * es) % 16 || ((long)p.ldb * es) % 16)
            return TC_ERR_ALIGN;
        if ((long)p.K * p.lda * es >= (1L << 32) || (long)p.K * p.ldb * es >= (1L << 32)) return TC_ERR_ARG;
        p.nk = p.K / kt;
        p.unit0 = (int)units;
        units += (long)(p.M / 128) * (p.N / 128) * p.nk;
        if (units >= (1L << 31)) return TC_ERR_ARG;
        g.p[i] = p;
    }
    static tc_dev_state dev_state;
    const int n_cu = tc_device_once(dev_state, [](int) {
        hipError_t a = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_grouped_kernel<MmaBF16>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 128 * TC_ROWB);
        hipError_t b = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_grouped_kernel<MmaF32>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 2 * 128 * TC_ROWB);
        return a != hipSuccess ? a : b;
    });
    if (n_cu < 0) return n_cu;
    g.total_units = (int)units;
    // one workgroup per CU (128 KB of LDS each); shares of at least 4 k-tiles so that tiny groups do not pay a prologue per unit
    long per = (units + n_cu - 1) / n_cu;
    if (per < 4) per = 4;
    g.units_per_wg = (int)per;
    // chunk of the contraction a tile is accumulated over before it is added = one share: a workgroup then flushes one or two
    // tiles, and the 32 workgroups of an XCD sit on 32 neighbouring tiles of the same token rows.  Measured on a decoder
    // layer's eight gradients (14 400 tokens, 155 k-tiles per share): chunk = share 126 us, half a share 131 us, a quarter
    // 147 us (more atomics), the whole contraction (tile-major, no sharing in L2) 196 us.
    // (chunk != share makes every share straddle two segments: a lone 512 x 512 problem 85 us instead of 27)
    long kc = per;
    g.kc = (int)kc;
    const unsigned grid = (unsigned)((units + per - 1) / per);
    if (dtype == TC_DTYPE_BF16)
        hipLaunchKernelGGL(gemm_tn_grouped_kernel<MmaBF16>, dim3(grid), dim3(256), 4 * 2 * 128 * TC_ROWB, stream, g);
    else
        hipLaunchKernelGGL(gemm_tn_grouped_kernel<MmaF32>, dim3(grid), dim3(256), 4 * 2 * 128 * TC_ROWB, stream, g);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

extern "C" int tcdiff_gemm_rowln(int dtype, const void* A, const void* W, int M, int K, int lda, int ldw, int a_mod,
                                 const tcdiff_row_epi* epi, hipStream_t stream) {
    if (!A || !W || !epi || M <= 0 || K <= 0) return TC_ERR_ARG;
    if (dtype != TC_DTYPE_BF16 && dtype != TC_DTYPE_F32 && dtype != TC_DTYPE_BF16X3) return TC_ERR_ARG;
    const int es = dtype == TC_DTYPE_BF16 ? 2 : 4;
    const int kt = dtype == TC_DTYPE_BF16 ? 64 : 32;
    if (K % kt != 0) return TC_ERR_ARG;
    if (!aligned16(A) || !aligned16(W) || ((long)lda * es) % 16 || ((long)ldw * es) % 16) return TC_ERR_ALIGN;
    // staging addresses are 32-bit offsets from the operand base
    if ((long)(a_mod > 0 ? a_mod : M) * lda * es >= (1L << 32) || (long)512 * ldw * es >= (1L << 32)) return TC_ERR_ARG;
    const int f = epi->flags;
    if ((f & TC_ROW_BIAS) && !epi->bias) return TC_ERR_ARG;
    if ((f & TC_ROW_LN_POST) && (!epi->ln_g || !epi->ln_b)) return TC_ERR_ARG;
    if ((f & TC_ROW_FILM) && (!epi->film || epi->film_ld % 4)) return TC_ERR_ARG;
    if ((f & (TC_ROW_FILM | TC_ROW_RES)) && !epi->xres) return TC_ERR_ARG;
    if ((f & TC_ROW_STORE_X) && !epi->xout) return TC_ERR_ARG;
    if ((f & TC_ROW_NEXT_LN) && (!epi->nln_g || !epi->nln_b)) return TC_ERR_ARG;
    if ((f & TC_ROW_STORE_H) && !epi->hout) return TC_ERR_ARG;
    if ((f & TC_ROW_STORE_ROT) && (!epi->rout || !epi->rope || !(f & TC_ROW_NEXT_LN))) return TC_ERR_ARG;
    if (epi->L <= 0) return TC_ERR_ARG;
    const void* vec16[] = {epi->bias, epi->ln_g, epi->ln_b, epi->film, epi->xres, epi->xout, epi->nln_g, epi->nln_b,
                           epi->hout, epi->rout, epi->rope};
    for (const void* p : vec16)
        if (p && !aligned16(p)) return TC_ERR_ALIGN;
    tcdiff_row_epi e = *epi;
    if (e.out_mul <= 0) e.out_mul = 1;
    if (e.groups <= 0) e.groups = 1;
    static tc_dev_state dev_state;
    const int n_cu = tc_device_once(dev_state, [](int) {
        hipError_t a = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rowln_kernel<MmaBF16>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, ROWLN_SMEM);
        hipError_t b = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rowln_kernel<MmaF32>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, ROWLN_SMEM);
        hipError_t c = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rowln_kernel<MmaBF16x3>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, ROWLN_SMEM);
        return a != hipSuccess ? a : b != hipSuccess ? b : c;
    });
    if (n_cu < 0) return n_cu;
    dim3 grid(((M + 63) / 64) * e.groups);
    if (dtype == TC_DTYPE_BF16)
        hipLaunchKernelGGL(gemm_rowln_kernel<MmaBF16>, grid, dim3(512), ROWLN_SMEM, stream, (const char*)A,
                           (const char*)W, M, K, (long)lda * es, (long)ldw * es, a_mod, e);
    else if (dtype == TC_DTYPE_BF16X3)
        hipLaunchKernelGGL(gemm_rowln_kernel<MmaBF16x3>, grid, dim3(512), ROWLN_SMEM, stream, (const char*)A,
                           (const char*)W, M, K, (long)lda * es, (long)ldw * es, a_mod, e);
    else
        hipLaunchKernelGGL(gemm_rowln_kernel<MmaF32>, grid, dim3(512), ROWLN_SMEM, stream, (const char*)A,
                           (const char*)W, M, K, (long)lda * es, (long)ldw * es, a_mod, e);
    TC_CHECK_LAUNCH();
    return TC_OK;
}
