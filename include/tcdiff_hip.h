/* tcdiff_hip.h -- C ABI of libtcdiff_gfx950.so: the MI355X (gfx950) kernels behind TCDiff's denoising hot path.
 *
 * The reference (Da1yuqin/TCDiff) has no FFI: its boundary for this path is the Python class surface
 * `model.model.DanceDecoder` / `model.diffusion.GaussianDiffusion` (TCDiff.py:18-19,76-102).  The host side
 * that mirrors that surface is `tcdiff_amd/{model,diffusion}.py`; every device operation it performs goes
 * through the entry points below (ctypes binding: tcdiff_amd/_lib.py).
 *
 * Conventions
 *   - every function returns 0 (TCDIFF_OK) or a negative error code; nothing is allocated, nothing
 *     synchronises the host; all pointers are caller-owned DEVICE pointers; work is enqueued on `stream`.
 *   - `dtype` selects the arithmetic of GEMM/attention operands: TC_DTYPE_BF16 (bf16 operands,
 *     v_mfma_f32_32x32x16_bf16, fp32 accumulate) or TC_DTYPE_F32 (fp32 operands, v_mfma_f32_32x32x2_f32:
 *     an exact fp32 fma chain -- the parity mode).  "T" below means that element type.  TC_DTYPE_BF16X3 (three GEMM-like
 *     launchers only, see its definition) keeps TC_DTYPE_F32's storage and evaluates products as split-bf16 triples.
 *   - the residual stream, LayerNorm statistics, softmax, FiLM and the diffusion update are always fp32.
 *   - matrices are row-major; `ld*` are leading dimensions in ELEMENTS; operand rows must be 16-byte
 *     aligned and K must be a multiple of 64 (bf16) / 32 (f32) elements (pad with zeros).
 */
#ifndef TCDIFF_HIP_H
#define TCDIFF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

#define TCDIFF_OK 0
#define TCDIFF_ERR_ARG (-1)
#define TCDIFF_ERR_ALIGN (-2)
#define TCDIFF_ERR_LAUNCH (-3)
#define TCDIFF_ERR_UNSUPPORTED (-4)

#define TC_DTYPE_F32 0
#define TC_DTYPE_BF16 1
#define TC_DTYPE_BF16X3 2 /* tcdiff_gemm_tile (forward epilogues), tcdiff_gemm_rowln, tcdiff_attention only: operands stored as fp32
                             exactly as for TC_DTYPE_F32; every product a b is formed as a_hi b_hi + a_hi b_lo + a_lo b_hi with
                             a_hi = bf16(a), a_lo = bf16(a - a_hi) -- three bf16 MFMAs, fp32 accumulate: ~2^-16 per product, the
                             reference's fp32 path (model/model.py:548-624 runs without autocast) to well inside 1e-3 */

/* activations (fused into GEMM epilogues and elementwise helpers) */
#define TC_ACT_NONE 0
#define TC_ACT_RELU 1
#define TC_ACT_GELU 2 /* exact erf form: F.gelu, TCDiff.py:85 */
#define TC_ACT_MISH 3 /* nn.Mish, model/model.py:157,457 */
#define TC_ACT_SILU 4 /* nn.SiLU, model/model.py:499 */

/* ---- gemm_tile epilogues ------------------------------------------------------------------------ */
#define TC_EPI_STORE_T 0   /* out[m][n] = T(act(acc + bias[n]))                                       */
#define TC_EPI_STORE_F32 1 /* out[m][n] = act(acc + bias[n]) as fp32                                  */
#define TC_EPI_QKV_HEADS 2 /* scatter to the head-major Q / K / V images read by tcdiff_attention     */
#define TC_EPI_ATOMIC_F32 3 /* out[m][n] += acc (fp32 atomics; split-K): only through tcdiff_gemm_splitk */

typedef struct {
    int mode;          /* TC_EPI_* */
    int act;           /* TC_ACT_* */
    float scale_q;     /* QKV: multiplies the Q columns (1/8 = 1/sqrt(d_k), model/model.py:97) */
    const float* bias; /* [N] or NULL */
    void* out;         /* STORE_T: T[M][ldc]; STORE_F32: float[M][ldc]; QKV: Q image */
    void* out_k;       /* QKV: K image    T[n_seq][H][Lp][64]                       */
    void* out_v;       /* QKV: V image    T[n_seq][H][Lp][64]                       */
    int ldc;
    int L, Lp, H;      /* QKV: tokens per sequence (row m -> sequence m / L, token m % L), padded length, heads */
    int n_q, n_k;      /* QKV: columns [0,n_q) are Q, [n_q,n_q+n_k) are K, the rest V (multiples of 128) */
    int tok_off;       /* QKV: added to the token index  */
    int seq_off;       /* QKV: added to the sequence index */
    int k_splits;      /* TC_EPI_ATOMIC_F32: workgroups per output tile (set by tcdiff_gemm_splitk; 0 elsewhere) */
    /* Training step, TC_EPI_STORE_T with act == NONE and N, ldc multiples of the 16-byte chunk: the activation (+ nn.Dropout)
     * that follows / precedes the nn.Linear in the same pass (model/model.py:244,399-400,490-494,522-528 and their autograd).
     *   out2 != NULL   : out = a = T(acc + bias) (kept for the backward), out2[m][n] = T(dropout(act2(a)))
     *   act_src != NULL: the GEMM is the input gradient of the NEXT linear, out = T(T(acc) * mask / (1 - p) * act2'(act_src[m][n]))
     * dropout: counter hash of (drop_seed (DEVICE int[2] or NULL), drop_site, m * N + n); drop_thr = 0 disables it. */
    void* out2; int ldc2;
    const void* act_src; int ld_src;
    int act2;
    const int* drop_seed; int drop_site; uint32_t drop_thr; float drop_scale;
    /* TC_EPI_QKV_HEADS: hgroup > 0 -- the columns of one of Q / K / V hold several images of hgroup heads each (all eight
     * decoder layers' cross-attention K, or V: model/model.py:394-396 evaluated for every layer in one GEMM); image g starts
     * hgroup_stride elements after image g - 1. */
    int hgroup; long hgroup_stride;
    /* small_m != 0: the caller's whole job is small -- tcdiff_gemm_tile may run the product as 32 x 32 tiles with K dealt to the
     * waves (gemm_small_kernel: bf16, TC_EPI_STORE_T / _F32 without out2 / act_src, N % 32 == 0, K % 32 == 0 <= 2048, when the
     * 128 x 128 tiling would leave three quarters of the chip idle).  Another summation order than the 128 x 128 tiling, hence opt-in:
     * with 0 an output element does not depend on how many rows share the launch. */
    int small_m;
} tcdiff_tile_epi;

/* C[M,N] = A[M,K] * W[N,K]^T with epilogue.  If A2 != NULL, output columns >= split_n (a multiple of 128)
 * take their A operand from A2 (same shape/ld as A): one launch computes Q,K from rot(h) and V from h
 * (model/model.py:374-383).  a_mod > 0: A row = m % a_mod.  An operand (rows x ld x element size) must span less than
 * 4 GB: tiles are staged with 32-bit offsets from the operand base (TC_ERR_ARG otherwise; same for tcdiff_gemm_rowln).
 * Replaces: nn.Linear calls model/model.py:78-80,399,454-465,490-494,522-528,560,623,164-166. */
int tcdiff_gemm_tile(int dtype, const void* A, const void* A2, int split_n, const void* W, int M, int N, int K,
                     int lda, int ldw, int a_mod, const tcdiff_tile_epi* epi, hipStream_t stream);

/* ---- gemm_rowln: N = 512, row-complete epilogue ------------------------------------------------- */
#define TC_ROW_BIAS 1       /* v = acc + bias[n]                                                       */
#define TC_ROW_LN_POST 2    /* v = LayerNorm_{ln_eps}(v) * ln_g + ln_b      (SBI_MSA.layer_norm, model/model.py:106) */
#define TC_ROW_FILM 4       /* v = (film[seq][n] + 1) * v + film[seq][512+n] (featurewise_affine, model/model.py:171) */
#define TC_ROW_RES 8        /* v = xres[m][n] + v  (implied by FILM as well)                           */
#define TC_ROW_STORE_X 16   /* xout[m'][n] = v (fp32)                                                  */
#define TC_ROW_NEXT_LN 32   /* u = LayerNorm_{nln_eps}(v) * nln_g + nln_b  (the next block's norm, model/model.py:326,332,338,344) */
#define TC_ROW_STORE_H 64   /* hout[m'][n] = T(u)   (without NEXT_LN: hout[m'][n] = T(v))                      */
#define TC_ROW_STORE_ROT 128 /* rout[m'][n] = T(rotary(u, pos = m' % L))   (model/model.py:375,387)    */

typedef struct {
    int flags;
    const float* bias;
    const float* ln_g;
    const float* ln_b;
    float ln_eps;
    const float* film; /* base of this block's FiLM rows: film[seq * film_ld + n] scale, +512 shift */
    int film_ld;
    const float* xres; /* fp32 [*,512] */
    int xres_mod;      /* > 0: residual row = m % xres_mod */
    float* xout;
    int L;             /* tokens per sequence */
    const float* nln_g;
    const float* nln_b;
    float nln_eps;
    void* hout;
    void* rout;
    const float* rope; /* [Lmax][512]: rope[p][2j] = cos(p*freq_j), rope[p][2j+1] = sin(p*freq_j) */
    int out_mul, out_add; /* output row m' = m * out_mul + out_add (0 -> 1) */
    int groups;           /* > 1: `groups` GEMMs of the same A in one launch; group g takes weight rows
                             [512 g, 512 g + 512), bias[512 g ..] and out_add + g (the per-dancer slices of the last
                             fusion-projection linear, model/model.py:527-528,561) */
} tcdiff_row_epi;

/* out rows[M,512] = epilogue(A[M,K] * W[512,K]^T).  Replaces fc+layer_norm+FiLM+residual
 * (model/model.py:103-106,327,334), linear2+FiLM+residual (:339,399-401), linear3(norm4(x)) (:344) and the
 * last fusion-projection linear (:527), each fused with the LayerNorm(+rotary) that consumes its result. */
int tcdiff_gemm_rowln(int dtype, const void* A, const void* W, int M, int K, int lda, int ldw, int a_mod,
                      const tcdiff_row_epi* epi, hipStream_t stream);

/* ---- training-side rows (forward pieces; csrc/train.hip) -----------------------------------------------------------
 * x_noisy[b][s*dn + d][c] = sqrt_ac[t_b] * x_start[b][d][s][c] + sqrt_1mac[t_b] * noise[b][s][d][c], channels 4 and 5
 * (trajectory) copied from x_start: q_sample + the restore + the (b, dn, S, C) -> (b, S*dn, C) permute of
 * model/diffusion.py:625-634,640-651.  t: int64 [b] on the device. */
int tcdiff_q_sample_traj(const float* x_start, const float* noise, const long* t, const float* sqrt_ac,
                         const float* sqrt_1mac, float* x_noisy, int b, int dn, int S, int C, hipStream_t stream);

/* axis_angle[i * per_row + j][0..2] = matrix_to_axis_angle(rotation_6d_to_matrix(rot6d + i * row_stride + 6 j))
 * (dataset/quaternion.py:28-32; pytorch3d 0.7.1 definitions restated, see oracle/tcdiff_oracle.py): per_row rotations
 * packed in every row of a matrix with `row_stride` floats per row (the 24 x 6 tail of a 151-wide motion row). */
int tcdiff_ax_from_6v(const float* rot6d, long n_rows, int per_row, long row_stride, float* axis_angle,
                      hipStream_t stream);

/* SMPLSkeleton.forward (vis.py:358-406): axis_angle [n][24][3], root [n][3] -> joints [n][24][3].  parents (HOST int[24],
 * a parent precedes its children) and offsets (HOST float[24][3]) are vis.py:48-101's constants or a caller's skeleton. */
int tcdiff_smpl_fk(const float* axis_angle, const float* root, long n, const int* parents, const float* offsets,
                   float* joints, hipStream_t stream);

/* out[b][4] = per-clip means of the reconstruction, velocity, FK and foot-skate terms of model/diffusion.py:668-733
 * (the first three times p2_weight[t_b]); model_out [b][S*dn][C], x_start in the dataset layout [b][dn][S][C],
 * joints_* [b*S*dn][24][3]; l1 != 0 selects F.l1_loss, else F.mse_loss. */
int tcdiff_loss_terms(const float* model_out, const float* x_start, const float* joints_model,
                      const float* joints_target, const float* p2_weight, const long* t, float* out, int b, int dn,
                      int S, int C, int l1, hipStream_t stream);

/* Adan.step (model/adan.py:33-123) over all parameter tensors in one launch: chunks of <= 65536 elements. */
typedef struct {
    float* p;          /* parameter */
    const float* g;    /* gradient */
    float* m;
    float* v;
    float* n_;
    float* pg;         /* prev_grad */
    long n;
} tcdiff_adan_chunk;

typedef struct {       /* Python-side doubles of adan.py rounded to fp32 where they meet a tensor */
    float b1, omb1, b2, omb2, b3, omb3;   /* betas and 1 - beta */
    float cm, cv, cn;                     /* bias corrections 1 / (1 - (1 - beta)^step) */
    float eps, lr, denom;                 /* denom = 1 + weight_decay * lr */
    int first;                            /* bit 0: step == 0 before the call: m, v, n are left untouched (adan.py:71);
                                             bit 1: RESTART of these tensors (adan.py:109-114): m = g, v = 0, n = g^2, then the
                                             parameter update; bit 2: prev_grad is not written (the caller evaluates
                                             restart_cond(state) on the old prev_grad between the two launches) */
} tcdiff_adan_scalars;

int tcdiff_adan_step(const tcdiff_adan_chunk* chunks, int n_chunks, const tcdiff_adan_scalars* scalars,
                     hipStream_t stream);

/* ---- row-block chains (bf16): everything between the two attentions of a decoder layer in ONE launch -------------
 * A block of 64 token rows stays on a CU; only weights stream (tcdiff_amd/csrc/chain.hip).
 *   TC_CHAIN_A      : O_self --fc, LayerNorm(1e-6), FiLM, +x--> x ; norm2, rotary ; w_qs --> Q image (cross-attention)
 *                     replaces model/model.py:103-106,327 and :332,387,78 (+ the 1/sqrt(d_k) of :97)
 *   TC_CHAIN_B      : O_cross --fc, LN, FiLM, +x--> x ; norm3 ; linear1, GELU ; linear2, FiLM, +x ; norm4 ; linear3 --> x' ;
 *                     next layer's norm1 (+rotary) ; w_qs / w_ks / w_vs --> Q, K, V images
 *                     replaces model/model.py:103-106,334,338-339,344,399-401 and the next layer's :326,374-383,78-80
 *   TC_CHAIN_B_LAST : the same up to linear3, whose bf16 rows feed the final projection (model/model.py:623)
 * `wstream`: the chain's weights as ONE linear stream per wave in consumption order, [8 waves][n_stages][4096 B]; a
 * stage is the v_mfma_f32_16x16x32_bf16 A-fragment image of one 32-deep k-step of the 64 (512-wide GEMMs: [n-tile 4][lane
 * group g 4][16 weight rows][16 B], lane group g holding k = 32 ks + 8 PI[g] .. +7 with PI = (0, 3, 1, 2)) or of two k-steps of
 * the 32 (linear1 chunk: [k-step 2][n-tile 2][g][16 rows][16 B]) weight rows wave w consumes; order: fc (16 stages), then for
 * chain A w_qs (16); for chain B four times {linear1 rows 256c+32w.. (8 stages), linear2 k-slice 256c.. (8)}, then
 * linear3 (16), w_qs, w_ks, w_vs of the NEXT layer (16 each).  n_stages = 32 / 144 / 96.  tcdiff_amd/engine.py packs it.
 * Rows: M token rows, L tokens per sequence; FiLM row = m / L, rotary position = m % L; head-major images as
 * TC_EPI_QKV_HEADS (H must be 8).  a_mod / xres_mod > 0: input / residual row = m % mod (layer 0 shares them between
 * the CFG branches).  The fp32 residual stream between chain launches is column-blocked (see the struct).
 *   TC_CHAIN_FULL / TC_CHAIN_FULL_LAST : chain A, then the CROSS-ATTENTION itself (head w on wave w, K / V from the
 *                     fragment-ordered cache images written by tcdiff_pack_kv_frags), then chain B / B_LAST, in one launch:
 *                     per layer the step is  self-attention -> one chain launch.  Stream: chain A's stages followed by
 *                     chain B's (176 / 128 stages); the first fc block uses film, n2_*, the second filmb,
 *                     n3_*; xres / xres_mod feed the first, xout carries x in between. */
#define TC_CHAIN_A 0
#define TC_CHAIN_B 1
#define TC_CHAIN_B_LAST 2
#define TC_CHAIN_FULL 3
#define TC_CHAIN_FULL_LAST 4
#define TC_CHAIN_FRONT 5 /* last fusion linear of one dancer over a block of 64 FRAMES (A = bf16 [M frames][1024], weights
                            rows [512 d, 512 d + 512) of relative_projection_layer.4, K = 1024: 32 stages) -> layer 0's
                            residual input xout (token row = frame dn + d, column-blocked with M dn rows); then layer 0's
                            norm1 (nn_g, nn_b) + rotary and w_qs / w_ks / w_vs (16 stages each) -> Q, K, V images.  M =
                            frames, L = TOKENS per sequence, b3 = all 512 dn biases; wstream = [dn][8 waves][80 stages];
                            grid = blocks x dn.  Replaces model/model.py:526-528,561 and layer 0's :326,374-383,78-80. */

typedef struct {
    int mode, n_stages;
    int M, L, a_mod, xres_mod, H, Lp;
    const void* A;        /* bf16 [*,512]: attention output rows */
    const void* wstream;
    const float* film;    /* FiLM of the attention block, PRE-FOLDED with the post-LayerNorm (SBI_MSA.layer_norm, eps ln_eps)
                             weights g, b in front of it: film[seq * film_ld + n] = g (scale + 1), +512: b (scale + 1) + shift,
                             so that the block's epilogue is x += LN(z) * G + Bv (model/model.py:103-106,171-173,327,334).  The
                             caller folds g, b into the DenseFiLM generator's weights (both rows are linear in its input):
                             tcdiff_amd/engine.py load_weights */
    const float* xres;    /* fp32 residual in: column-blocked (below) with M (xres_mod > 0: xres_mod) rows, or, with
                             xres_rowmajor, plain [*,512] rows (layer 0: written by tcdiff_gemm_rowln) */
    float* xout;          /* fp32 residual out, column-blocked [64][M][8] (chain B: holds x between the blocks, then x') */
    const float* n2_g;    /* the norm that follows: norm2 (chain A) / norm3 (chain B) */
    const float* n2_b;
    const float* rope;    /* cos/sin table of tcdiff_rope_table, column-blocked [64][rope_rows][8] */
    void* q_out;          /* Q image T[n_seq][8][Lp][64] (chain A: cross-attention Q; chain B: next layer's) */
    const float* b1;      /* linear1 bias [1024] */
    const float* film3;   /* FiLM of the feed-forward block, pre-folded with linear2's bias b2: [scale + 1 | b2 (scale + 1) + shift]
                             (model/model.py:339,399-401) */
    const float* n4_g;
    const float* n4_b;
    const float* b3;      /* linear3 bias */
    const float* nn_g;    /* next layer's norm1 */
    const float* nn_b;
    void* k_out;
    void* v_out;
    void* h_out;          /* *_LAST modes: see out_ld */
    int film_ld;
    float ln_eps, n2_eps, n4_eps, nn_eps, scale_q;
    /* TC_CHAIN_FULL*: the cross-attention block */
    const float* filmb;   /* FiLM of the cross-attention block, pre-folded with multihead_attn.layer_norm like `film` */
    const float* n3_g;    /* norm3 */
    const float* n3_b;
    const void* kf;       /* fragment-ordered K cache of this layer: T[n_kv][8 heads][nkt][4][64 lanes][8] */
    const void* vf;       /* fragment-ordered V cache                                                      */
    int n_shared, nkt, Lk; /* kv slot = seq < n_shared ? 0 : seq - n_shared + (n_shared > 0); nkt = ceil(Lk / 32) tiles */
    /* COLUMN-BLOCKED fp32 [rows][512] matrix: element (row, c) at ((c / 8) * rows + row) * 8 + c % 8 -- the 32 rows a
     * wave instruction touches are then one contiguous kilobyte instead of 32 separate 32-byte pieces */
    int xres_rowmajor;     /* 1: xres is a plain row-major [*,512] matrix */
    int rope_rows;         /* rows of the column-blocked rotary table (>= L) */
    int dn;                /* TC_CHAIN_FRONT: dancers */
    int mt;                /* 16-row tiles per row block: 0 = chosen from M and the device's CU count (4 = 64-row blocks when that
                              fills the chip, else 2 or 1: a small job runs on many small blocks), or 1 / 2 / 4 (tests) */
    int out_ld;            /* *_LAST modes: 0 -> h_out = bf16 [M,512] rows of linear3; > 0 -> h_out = fp32 [M][out_ld], the
                              first out_ld columns of linear3 (the caller folded final_layer into its weights and bias:
                              model/model.py:344,623); out_ld % 4 == 0 */
    int nw;                /* waves per workgroup: 0 or 8 = eight waves x 64 output columns; 4 = four waves x 128 columns, one per
                              SIMD (TC_CHAIN_FULL, _FULL_LAST, _FRONT only).  `wstream` must be packed for the same form: per wave
                              stages of 16 nt x 64 lanes x 16 B with nt = 4 (4-KB stages) or 8 (8-KB stages) n-tiles */
    /* ---- round 5: the layer's SELF-attention inside the launch (TC_CHAIN_FULL / _FULL_LAST, nw = 8) ------------------------
     * seq_blocks = 1: row blocks are cut per sequence -- block (s, b) holds rows s L + 16 mt b .. of sequence s only (grid = (M / L)
     * x ceil(L / (16 mt)); no block straddles two sequences, its rows are keys 16 mt b .. of its sequence).
     * sa_q != NULL: instead of reading the attention output `A`, the block computes softmax((Q / 8) K^T) V of its rows itself
     * (model/model.py:97-102; wave = head, online softmax over 32-key tiles as the in-kernel cross-attention) from
     *   sa_q  : the block's Q^T fragments, bf16, as the PREVIOUS launch left them in `qf_out`: [block][8 waves][8 pieces = (row tile
     *           mt < 4, d-step s < 2)][64 lanes][8] -- 8 KB per (block, wave) whatever the launch's rows per block -- scaled by
     *           log2(e) / sqrt(d_k): the in-kernel softmax works in the exp2 domain;
     *   sa_kf, sa_vf : K / V of the layer in the fragment order of tcdiff_pack_kv_frags, bf16 [M / L][8 heads][sa_nkt][4][64][8],
     *           sa_nkt = ceil(L / 32) tiles; keys >= L of the last tile must be finite (the images must start zeroed: a launch
     *           writes the keys it owns, in 16-row blocks one 8-byte half of a V piece per lane).
     * qf_out / kf_out / vf_out != NULL (all three together; requires seq_blocks): the next layer's Q, K, V leave in those formats
     * (written straight from the accumulators: 1-KB pieces per wave instruction) instead of the head-major images q_out / k_out /
     * v_out; out_nkt = the key tiles per (sequence, head) of kf_out / vf_out. */
    int seq_blocks;
    const void* sa_q;
    const void* sa_kf;
    const void* sa_vf;
    int sa_nkt;
    void* qf_out;
    void* kf_out;
    void* vf_out;
    int out_nkt;
} tcdiff_chain_args;

int tcdiff_chain(const tcdiff_chain_args* args, hipStream_t stream);

/* The same decoder layer for SMALL jobs (tcdiff_amd/csrc/chain_split.hip; what TCDiff.py renders: a handful of clips,
 * TCDiff.py:292-303, model/diffusion.py:386-442): FOUR workgroups per 16-row block, each streaming about a third of the layer's
 * weights, as FOUR launches `part` = 1 .. 4 whose boundaries are the exchanges between the four --
 *   1: self-attention of two heads per workgroup (sa_q / sa_kf / sa_vf; without sa_q: the rows of `A`), fc split over K -> p_out
 *   2: sum of p_in; LN, FiLM, +xres -> xout; norm2, rotary; w_qs and cross-attention of two heads; fc split over K   -> p_out
 *   3: sum of p_in; LN, FiLM, +xres -> xout; norm3; linear1 rows 256 c .., GELU, linear2 over those 256              -> p_out
 *   4: sum of p_in; FiLM, +xres; norm4; linear3 (+ the folded final layer) -> xout / h_out; norm1, rotary; Q / K / V fragment images
 * args: as for TC_CHAIN_FULL / TC_CHAIN_FULL_LAST with seq_blocks = 1, nw = 0 / 8, the same 176 / 128-stage `wstream`; the layer's
 * fragment-order outputs (qf_out / kf_out / vf_out) are mandatory unless LAST.  `xres` -> `xout` is NOT in place (parts 2, 3, 4 read
 * whole rows and store quarters: the caller alternates two buffers); p_in / p_out: fp32 [blocks][4][16][512] partial sums, two
 * buffers alternating likewise.  grid = (M / L) * ceil(L / 16) * 4 workgroups: meant for jobs where that fits the chip.
 * part 1 with sa_q and a_mod > 0: sequence s reads the fragment images of sequence s % (a_mod / L) (layer 0 under classifier-free
 * guidance: the stacked branches share x).
 * part = 12: parts 1 and 2 as one launch (no p_in): every member computes the self-attention of all eight heads and the whole fc
 * (more K / V and weight bytes per member, one exchange less: for short sequences).
 * part = 0 (mode TC_CHAIN_FRONT, the 80-stage front stream of any dancer): layer 0's Q / K / V fragment images from the token
 * rows the fusion projection left -- xres: ROW-MAJOR fp32 [M][512] (model/model.py:561 as one [frames][512 dn] product, which is
 * the same memory); nn_g / nn_b / nn_eps: layer 0's norm1; rope; qf_out / kf_out / vf_out / out_nkt; scale_q.  With it a small job
 * runs the fusion projection as plain small products (tcdiff_gemm_tile picks its small-M kernel) and layer 0's self-attention
 * inside part 1, instead of the TC_CHAIN_FRONT launch (9 workgroups for one 3 x 150 clip) and a stand-alone attention. */
int tcdiff_chain_split(const tcdiff_chain_args* args, int part, const float* p_in, float* p_out, hipStream_t stream);

/* Fragment-ordered images of the cross-attention K / V caches for TC_CHAIN_FULL (bf16): for keys key_lo <= key < key_hi
 * of every (slot, head) of Kc / Vc (T[n_slots][H][Lp][64], natural [key][d] rows),
 *   Kf[slot][head][key / 32][(key % 32) / 16][d / 32][16 g + key % 16][jj]  with d % 32 = 16 (jj / 4) + 4 g + jj % 4
 *   Vf[slot][head][key / 32][d / 16][16 g + d % 16][jj]                     with key % 32 = 16 (jj / 4) + 4 g + jj % 4
 * i.e. each 1-KB piece is the 64 lanes' 16-byte v_mfma_f32_16x16x32_bf16 A-operand fragments of one (16-key tile, 32-deep
 * d-step) of K, or (16-d tile, 32-key step) of V^T, in the k order in which the chain kernel's accumulator tiles hold Q^T
 * and P^T (two 16-row tiles, rows 4 g + j of each).  Keys >= Lk inside
 * the last tile must hold finite values (zero).  Run once over [0, Lk) when a cache slot is filled and over the two
 * time-token keys every step. */
int tcdiff_pack_kv_frags(const void* Kc, const void* Vc, void* Kf, void* Vf, int n_slots, int H, int Lp, int nkt,
                         int key_lo, int key_hi, hipStream_t stream);

/* ---- fused attention ------------------------------------------------------------------------------
 * O[(seq*Lq + q)*ldo + head*64 + d] = softmax_k(Q[seq][head][q] . K[kv][head][k]) V[kv][head][k][d]
 * Q  : T[n_seq][H][Lp_q][64]   (already scaled by 1/sqrt(64))
 * K,V: T[n_kv ][H][Lp_k][64]   natural [key][feature] rows (written by TC_EPI_QKV_HEADS / tcdiff_scatter_time_kv)
 * kv = seq < n_shared ? 0 : seq - n_shared + (n_shared > 0)   (the unconditional CFG branch shares one K/V)
 * Lp_q % 128 == 0, Lp_k % 64 == 0, pad rows of Q/K/V must be finite (zero).  Keys >= Lk are masked.
 * ng: query rows per wave of the K/V-resident bf16 kernel in units of 32 (1 or 2); 0 = chosen from the launch size
 * and the device's CU count.  Results do not depend on ng beyond fp32 summation order inside a row.
 * Replaces model/model.py:97-102 (SBI_MSA core) and nn.MultiheadAttention's core (model/model.py:228-236). */
int tcdiff_attention(int dtype, const void* Q, const void* K, const void* V, void* O, int n_seq, int H, int Lq,
                     int Lk, int Lp_q, int Lp_k, int ldo, int n_shared, int ng, hipStream_t stream);

/* ---- LayerNorm (+ rotary) prologue: one wave per 512-wide row ---------------------------------------
 * u = LayerNorm_eps(x[row]) * g + b; optional outputs: h = T(u); rot = T(rotary(u, pos)); y32 = u (fp32).
 * pos = pos_base + (row % pos_mod).  Replaces model/model.py:326 etc. where no producing GEMM can fuse it,
 * norm_cond (:616) and the music encoder's norms (:220-221). */
int tcdiff_ln_rot(int dtype, const float* x, int rows, const float* g, const float* b, float eps, void* h,
                  void* rot, float* y32, const float* rope, int pos_mod, int pos_base, hipStream_t stream);

/* rope[p][2j] = cos(p * freqs[j]), rope[p][2j+1] = sin(p * freqs[j]), p < n_pos, j < 256
 * (model/rotary_embedding_torch.py:115-130 + :58: the table the reference recomputes on every call). */
int tcdiff_rope_table(const float* freqs, float* rope, int n_pos, hipStream_t stream);

/* ---- small elementwise helpers of the step-invariant / per-step conditioning path ----------------- */
/* dst[r][c] = T(src row r)[c] for c < cols, 0 for cols <= c < ld_dst.  Source row r lives at
 * src + (r / rows_per_batch) * batch_stride + (r % rows_per_batch) * row_stride  (strides in floats). */
int tcdiff_convert_pad(int dtype, const float* src, void* dst, int rows, int cols, int ld_dst, int rows_per_batch,
                       long batch_stride, long row_stride, hipStream_t stream);
/* emb[i] = T([sin(t_i * f_k), cos(t_i * f_k)]), k < 256 (SinusoidalPosEmb, model/utils.py:36-48); times int32 */
int tcdiff_sinusoidal(int dtype, const int* times, int n, const float* freq, void* emb, hipStream_t stream);
/* out[b][c] = mean_s x[b][s][c]   (model/model.py:593) */
int tcdiff_mean_pool(const float* x, float* out, int B, int S, int C, hipStream_t stream);
/* out[i][c] = T(act(a[ia[i]][c] + (b ? b[i][c] : 0))), c < 512; ia == NULL -> identity
 * (t = to_time_cond(t_hidden) + cond_hidden; Mish(t): model/model.py:612,157) */
int tcdiff_add_act(int dtype, const float* a, const int* ia, const float* b, int n, int act, void* out,
                   float* out32, hipStream_t stream);
/* per-step: copy the two time-token K/V rows of every layer into the cross-attention caches.
 * tab: T[NL][n_t][2][1024] (K cols 0..511, V cols 512..1023); tidx[seq] selects the row set.
 * Kc, Vc: T[NL][n_kv][H][Lp][64]; rows tok0, tok0+1.  */
int tcdiff_scatter_time_kv(int dtype, const void* tab, int n_t, const int* tidx, void* Kc, void* Vc, int NL,
                           int n_kv, int H, int Lp, int tok0, hipStream_t stream);

/* ---- sampler steps --------------------------------------------------------------------------------
 * step scalars live in DEVICE memory so that one captured hipGraph serves every step:
 *   counter[0] = index of the current step; params[step][8] floats, tseq[step] = timestep index.
 * tcdiff_step_begin: tidx[i] = tseq[counter[0]] for i < n.   tcdiff_step_end: counter[0] += 1.          */
int tcdiff_step_begin(const int* counter, const int* tseq, int* tidx, int n, hipStream_t stream);
int tcdiff_step_end(int* counter, hipStream_t stream);

/* tcdiff_step_prologue: step_begin + add_act(mish) + scatter_time_kv (+ its pack_kv_frags image) + convert_pad of x_t
 * + the counter bump of step_end in ONE launch (model/model.py:606-612 conditioning of a step; model/diffusion.py:245
 * loop variable).  counter is int[4] = {current step, seed0, seed1, next step}; the launch reads counter[3] and writes
 * counter[0] = that step; tcdiff_sampler_update(mode | TC_SAMPLER_ADVANCE) later writes counter[3] = counter[0] + 1,
 * so that no tcdiff_step_end launch is needed.
 * Kc/Vc (row-major caches) or Kf/Vf (fragment images, bf16 only) may be NULL, not both; x may be NULL (no copy); with
 * film_tab the FiLM generator input (film_in) is not produced -- the step needs no FiLM GEMM. */
typedef struct tcdiff_step_prologue_args {
    int* counter;
    const int* tseq;
    int* tidx;            /* [n_seq] <- timestep (kept for the op-by-op kernels) */
    const float* t_base;  /* [n_t][512] time-embedding MLP output per timestep */
    const float* hidden;  /* [n_seq][512] cond hidden rows */
    void* film_in;        /* [n_seq][512] model dtype */
    int n_seq;
    const void* tab;      /* [NL][n_t][2][1024] time-token K|V rows, model dtype */
    int n_t;
    void *Kc, *Vc, *Kf, *Vf;
    int NL, n_kv, H, Lp, nkt, tok0;
    const float* x;       /* [rows][nfeat] fp32 x_t or NULL */
    void* xin;            /* [rows][ld_xin] model dtype, zero padded */
    int rows, nfeat, ld_xin;
    /* FiLM rows from a per-job table instead of a per-step GEMM (NULL: off).  film_tab[n_t][film_rows][nfilm] fp32 holds
     * Linear(Mish(t_base[t] + hidden_j)) of all 24 DenseFiLM blocks (model/model.py:154-168,612) for every timestep row t and
     * every distinct conditioning row j (0 = the null conditioning shared by the unconditional branch, 1 + i = clip i);
     * film_out[seq] <- film_tab[t][seq < n_unc ? 0 : seq - n_unc + 1] for seq < n_seq. */
    const float* film_tab;
    float* film_out;
    int film_rows, nfilm, n_unc;
    /* 0: everything in one launch.  Round 6: the step's first GEMM needs only x_t, the first decoder-layer launch ~110 us later
     * needs the FiLM rows and the time-token rows -- so the captured step launches the prologue in two PARTS, the second on a
     * forked stream beside the input / fusion GEMMs (which leave ~100 CUs idle):
     * TC_PROLOGUE_X    = tidx, the model-dtype copy of x_t, counter[0] <- step;
     * TC_PROLOGUE_COND = the FiLM generator input or the gathered FiLM rows, the time-token K / V rows (it only READS counter[3]). */
    int parts;
} tcdiff_step_prologue_args;
#define TC_PROLOGUE_X 1
#define TC_PROLOGUE_COND 2
int tcdiff_step_prologue(int dtype, const tcdiff_step_prologue_args* a, hipStream_t stream);

/* DDPM: params = {w, coef1, coef2, sigma}  (model/diffusion.py:217-252, model/model.py:546)
 *   x0 = clamp(unc + (cond - unc) * w, -1, 1);  x <- coef1*x0 + coef2*x + sigma*eps
 * DDIM: params = {w, sqrt_recip_ac, sqrt_recipm1_ac, sqrt_ac_next, c, sigma, last}  (model/diffusion.py:195-204,407-431)
 *   x0 = clamp(...); pn = (sqrt_recip_ac*x - x0)/sqrt_recipm1_ac; x <- last ? x0 : sqrt_ac_next*x0 + c*pn + sigma*eps
 * out_unc / out_cond: fp32 [n_rows][ldo] network outputs (out_unc may be NULL when w == 1 is known: x0 = clamp(cond)).
 * eps: fp32 [n_rows][nfeat] or NULL -> Philox4x32-10 normal keyed by (seed, clip0 + row / L, timestep, element);
 *      counter[1], counter[2] are device-side seed words XORed into `seed` (re-seed a captured graph).
 * traj: optional fp32 [n_rows][3]: channels 4,5 of x are overwritten with traj[...,0:2] after the update
 * (model/diffusion.py:427-431).  x is updated in place; x0_out (optional) receives x0. */
#define TC_SAMPLER_DDPM 0
#define TC_SAMPLER_DDIM 1
#define TC_SAMPLER_ADVANCE 0x100 /* OR into mode: also write counter[3] = counter[0] + 1 (pairs with tcdiff_step_prologue) */
int tcdiff_sampler_update(int mode, const float* out_unc, const float* out_cond, int ldo, float* x,
                          const float* eps, const float* traj, float* x0_out, int n_rows, int nfeat, int L,
                          const int* counter, const float* params, const int* tseq, uint64_t seed, int clip0,
                          hipStream_t stream);

/* Constraints re-imposed after a sampler step, as a launch of the captured step (no host callback between steps).
 *   kind 1: x = mask ? value : x unless this is the last step        (ddim_sample_Footwork, model/diffusion.py:341-356)
 *   kind 2: x = (p4 * value + p5 * noise) * mask + (1 - mask) * x    (inpaint_loop's q_sample(value, t-1) blend, :545-551)
 * Bit 1 of (int)params[step][7] enables the step (the host writes "not last" / "t > 0"); columns 4, 5 hold sqrt_ac[t-1],
 * sqrt(1 - ac[t-1]) for kind 2 (DDPM steps, whose own scalars are columns 0..3).  mask: fp32 [mask_rows][nfeat] (row % mask_rows: one [L][nfeat] mask may serve every
 * clip); value, q_eps: fp32 [n_rows][nfeat]; q_eps NULL -> Philox normal (stream word 1, same keying as the step). */
int tcdiff_sampler_constrain(int kind, float* x, const float* mask, int mask_rows, const float* value,
                             const float* q_eps, int n_rows, int nfeat, int L, const int* counter, const float* params,
                             const int* tseq, uint64_t seed, int clip0, hipStream_t stream);

/* tcdiff_window_couple gated by bit 0 of (int)params[step][7] (the reference does not couple after the last step). */
int tcdiff_window_couple_step(float* x, int b, int seq_len, int row_elems, const int* counter, const float* params,
                              hipStream_t stream);

/* y[row][c] = unc + (cond - unc) * w, c < nfeat  (DanceDecoder.guided_forward, model/model.py:546) */
int tcdiff_cfg_combine(const float* out_unc, const float* out_cond, int ldo, float w, float* y, int n_rows,
                       int nfeat, hipStream_t stream);

/* x[1:, :half] = x[:-1, half:] on the (b, seq_len, dn*nfeat) view (model/diffusion.py:502-506,599-601) */
int tcdiff_window_couple(float* x, int b, int seq_len, int row_elems, hipStream_t stream);

/* ---- EMA of the master weights (training-side, model/diffusion.py:61-76 EMA.update_model_average) -------------
 * ma = ma * beta + (1 - beta) * cur over a list of fp32 tensors in ONE launch (the reference issues three elementwise
 * kernels per parameter tensor, 435 tensors).  `chunks` is a DEVICE array: chunk i covers chunks[i].n <= 65536 elements.
 * Rounding as torch's `old * beta + (1 - beta) * new`: two products rounded to fp32, then the sum (no fma). */
typedef struct {
    float* ma;
    const float* cur;
    long n;
} tcdiff_ema_chunk;
int tcdiff_ema_update(const tcdiff_ema_chunk* chunks, int n_chunks, float beta, float one_minus_beta,
                      hipStream_t stream);

/* =====================================================================================================================
 * Training step: train-mode forward pieces and the backward pass (csrc/train_ops.hip, attention_train.hip, gemm.hip).
 * The reference has no backward code: it is torch autograd over model/model.py + model/diffusion.py:636-741, driven by
 * accelerator.backward(total_loss) at TCDiff.py:232.  tcdiff_amd/train_engine.py sequences these launchers as an
 * explicit forward + reverse schedule behind ONE torch.autograd.Function.
 *
 * Gradient dtypes: the gradient of a T-typed activation is T; the gradient of the fp32 residual stream is fp32;
 * parameter gradients are fp32 and ACCUMULATE (+=) into caller-zeroed buffers.
 *
 * Dropout (train mode; model/model.py:98,103,240,244-245,383,396,400-401 and nn.MultiheadAttention's weights) is a
 * counter-based hash of (seed, site, flat element index) -- csrc/train_common.h -- regenerated by the backward kernels.
 * `seed` is a DEVICE int[2] (NULL = {0,0}); drop_thr = floor(p * 2^32) (0 disables), drop_scale = 1 / (1 - p). */
#define TC_SITE_ENC(i, k) (4 * (i) + (k))        /* encoder layer i: k = 0 attention weights, 1 dropout1, 2 inner, 3 dropout2 */
#define TC_SITE_DEC(l, k) (16 + 8 * (l) + (k))   /* decoder layer l: k = 0 self weights, 1 self fc, 2 dropout1, 3 cross
                                                    weights, 4 cross fc, 5 dropout2, 6 inner, 7 dropout3 */

/* dst[r][c] = T(src[r][c]) (c < cols; zeros for cols <= c < cols_pad) and/or dstT[c][r] = T(src[r][c]) (r < rows; zeros
 * for rows <= r < rows_pad), c < cols.  src is fp32 (src_f32 != 0) or T.  colsum (optional, fp32 [cols], caller-zeroed)
 * += sum_r src[r][c]: the bias gradient of an nn.Linear while its dY is being repacked.  Produces the K-contiguous
 * operands tcdiff_gemm_tile needs for dgrad (W^T) and wgrad (dY^T, X^T). */
int tcdiff_cast_transpose(int dtype, int src_f32, const void* src, int rows, int cols, int ld_src, void* dst, int ld_dst,
                          int cols_pad, void* dstT, int ld_dstT, int rows_pad, float* colsum, hipStream_t stream);

/* The same for a table of fp32 matrices in ONE launch -- the operand packs (W as T [N, Kp] and W^T as T [K, Np]) of every
 * nn.Linear, rebuilt from the fp32 master parameters after each optimizer step (the reference's autocast does the
 * equivalent cast per use: accelerate mixed_precision, TCDiff.py:96-104).  `descs_dev` is a DEVICE array; fill each entry's
 * src .. rows_pad, call tcdiff_ct_desc_init (host) for tiles_x / vec and the tile count, and set tile0 to the running sum
 * of the counts; n_tiles = the total. */
typedef struct tcdiff_ct_desc {
    const void* src;                  /* fp32 [rows][ld_src] */
    void* dst;                        /* T [rows][ld_dst], columns cols .. cols_pad zero-filled; or NULL */
    void* dstT;                       /* T [cols][ld_dstT], columns rows .. rows_pad zero-filled; or NULL */
    int rows, cols, ld_src, ld_dst, cols_pad, ld_dstT, rows_pad;
    int tile0, tiles_x, vec;          /* tile0: caller; tiles_x, vec: tcdiff_ct_desc_init */
} tcdiff_ct_desc;
int tcdiff_ct_desc_init(int dtype, tcdiff_ct_desc* d);
int tcdiff_cast_transpose_multi(int dtype, const tcdiff_ct_desc* descs_dev, int n_desc, int n_tiles, hipStream_t stream);

/* ---- row-block GEMM of the training step (bf16; csrc/gemm_rows.hip) -------------------------------------------------
 * C[M, N] = A[M, K] * Wn[N, K]^T with tcdiff_gemm_tile's epilogues, for the tall products inside the decoder layers
 * (K = 512 or 1024, N a multiple of 512): a workgroup keeps 64 / 32 / 16 rows of A in LDS (mt = 4 / 2 / 1; 0 = by M and the
 * CU count) and streams the weights, which arrive as per-wave fragment streams -- the 4-KB stage format of tcdiff_chain:
 * wstream = [8 waves][N / 512 phases x K / 32 stages][4096 B], wave w's stage (p, ks) holding [n-tile 4][lane group 4][16 rows]
 * [8 k] = Wn[512 p + 64 w + 16 nt + c][32 ks + 8 PI(g) + j], PI = (0, 3, 1, 2).  tcdiff_pack_row_streams builds such streams on
 * the device from fp32 matrices: Wn[n][k] = src[n sn + k sk], i.e. (sn, sk) = (ld, 1) for the forward product of an nn.Linear
 * weight [N, K] and (1, ld) for its input gradient (Wn = W^T); an entry may be a piece of the packed matrix (stacked
 * parameters).  descs_dev: DEVICE array; max_elems = the largest N * K in it.
 * A2 / split_n as in tcdiff_gemm_tile (K = 512 only; split_n a multiple of 512).  Epilogue restrictions: act == TC_ACT_NONE,
 * hgroup == 0; TC_EPI_QKV_HEADS: H = 8, n_q and n_k are 0 or 512 and at most 512 columns of V (one phase per image), no
 * tok_off / seq_off, L >= 64; ldc / ldc2 / ld_src multiples of 8 (fp32 output: of 4).  Returns TC_ERR_UNSUPPORTED for shapes it
 * does not take (the caller then uses tcdiff_gemm_tile).
 * Replaces in the training step: nn.Linear forward and input-gradient products of model/model.py:78-80,103,399-401,344. */
typedef struct tcdiff_ws_desc {
    const float* src;
    void* dst;
    long sn, sk;                      /* element strides of Wn's n and k in src */
    int N, K;                         /* this piece: N % 512 == 0, K % 32 == 0 */
    int np_dst, p0, kst_dst, ks0;     /* dst is a stream of np_dst phases x kst_dst stages per wave; the piece's phase p, k-step ks
                                         land at (p0 + p, ks0 + ks).  A whole matrix: np_dst = N / 512, kst_dst = K / 32, p0 = ks0 = 0 */
} tcdiff_ws_desc;
int tcdiff_pack_row_streams(const tcdiff_ws_desc* descs_dev, int n_desc, int max_elems, hipStream_t stream);
int tcdiff_gemm_rows(const void* A, const void* A2, int split_n, const void* wstream, int M, int N, int K, int lda,
                     const tcdiff_tile_epi* epi, int mt, hipStream_t stream);

/* Split-K GEMM with fp32 accumulation into `out`: out[m][n] += sum_k A[m][k] W[n][k] (atomic adds; `splits` workgroups
 * share each 128x128 tile).  The weight gradient dW[N,K] += dY^T X of every nn.Linear, with M = out features, N = in
 * features and the contraction over the token rows. */
int tcdiff_gemm_splitk(int dtype, const void* A, const void* W, int M, int N, int K, int lda, int ldw, float* out,
                       int ldc, int splits, hipStream_t stream);

/* The weight gradient without transposed copies: out[m][n] += sum_k A[k][m] B[k][n], A = dY [tokens][lda], B = X
 * [tokens][ldb] as the forward left them (token-major).  M, N multiples of 128, K (tokens) a multiple of the k-tile (64
 * bf16 / 32 f32), else TC_ERR_UNSUPPORTED and the caller goes through tcdiff_cast_transpose + tcdiff_gemm_splitk.
 * Autograd of nn.Linear's weight in the reference (torch addmm backward; model/model.py:78-80,103,399-401). */
int tcdiff_gemm_tn(int dtype, const void* A, const void* B, int M, int N, int K, int lda, int ldb, float* out, int ldc,
                   int splits, hipStream_t stream);

/* Several weight gradients in ONE launch: out_i[m][n] += sum_k A_i[k][m] B_i[k][n] for i < n_prob <= TC_TN_MAX_PROB, each
 * with tcdiff_gemm_tn's shape rules.  The (tile, k-tile) work units of all problems are cut into equal contiguous shares,
 * one workgroup per CU: every CU runs the same number of k-tiles and a tile is added (fp32 atomics) once per share that
 * touches it.  `probs` is a HOST array (copied into the kernel arguments); fill A .. ldc, the rest is the launcher's.
 * The backward of a decoder layer queues its seven weight gradients and flushes them here (train_engine.py). */
#define TC_TN_MAX_PROB 16
typedef struct {
    const void* A; const void* B; float* out;      /* A [K][lda] (dY), B [K][ldb] (X), out [M][ldc] fp32 */
    int M, N, K, lda, ldb, ldc;
    int nk, unit0;                                  /* set by tcdiff_gemm_tn_grouped */
} tcdiff_tn_problem;
typedef struct {
    int n_prob, total_units, units_per_wg, kc;      /* kc: k-tiles per chunk of the unit order */
    tcdiff_tn_problem p[TC_TN_MAX_PROB];
} tcdiff_tn_group;
int tcdiff_gemm_tn_grouped(int dtype, const tcdiff_tn_problem* probs, int n_prob, hipStream_t stream);

/* x[m][c] = dropout(x[m][c] + pe[m % pos_mod][c]) in place on fp32 rows [rows][cols] (cols % 4 == 0); pe == NULL: dropout only;
 * drop_thr == 0: the addition only.  Hash index = m * cols + c (the index tcdiff_act_drop_bwd uses, which carries the backward).
 * Replaces PositionalEncoding.forward in train mode, model/utils.py:27-32 as called at model/model.py:564,580 (use_rotary=False). */
int tcdiff_pos_drop(float* x, int rows, int cols, const float* pe, int pos_mod, const int* seed, int site, uint32_t drop_thr,
                    float drop_scale, hipStream_t stream);

/* y = T(dropout(act(a)))  /  da = dy * mask / (1 - p) * act'(a).  a, da: fp32 (a_f32 != 0) or T [rows][ld_a]; y, dy:
 * T [rows][ld_y]; columns >= cols of y / da are written as zeros up to the leading dimension.  Hash index = r * cols + c.
 * Replaces the activations + nn.Dropout of model/model.py:244,400 (GELU), :490-494,522-528 (ReLU), :454-458,157 (Mish),
 * :496-501 (SiLU) and their autograd. */
int tcdiff_act_drop(int dtype, int a_f32, const void* a, int ld_a, void* y, int ld_y, int rows, int cols, int act,
                    const int* seed, int site, uint32_t drop_thr, float drop_scale, hipStream_t stream);
int tcdiff_act_drop_bwd(int dtype, int a_f32, const void* a, int ld_a, const void* dy, int ld_y, void* da, int rows,
                        int cols, int act, const int* seed, int site, uint32_t drop_thr, float drop_scale,
                        hipStream_t stream);

/* The row-local glue between two GEMMs of a transformer block, forward and backward, one wave per 512-wide row:
 *   u = dropout_pre(z + bias);  y = dropout_post(LayerNorm_post(u));  v = (scale + 1) y + shift;  xn = xres + v;
 *   hn = LayerNorm_next(xn);  rot = rotary(hn, pos)                      (each stage optional: TC_ROWF_* flags)
 * = model/model.py:103-106 (fc dropout + layer_norm), :327,334,339 (dropout1..3 + featurewise_affine + residual),
 *   :326,332,338,344 (the next norm), :375,387-388 (rotary); encoder :220-221,240,245.
 * Backward (tcdiff_row_bwd) recomputes the forward from (z, xres) and returns d_z (T), d_xres (fp32), per-sequence FiLM
 * gradients (atomic += into d_film) and the LayerNorm / bias gradients -- either per-block partial sums in `partials`
 * ([grid blocks][5][512] fp32: d_bias, d_ln_g, d_ln_b, d_nln_g, d_nln_b), folded by tcdiff_row_param_reduce in a fixed
 * order, or atomic += into g_bias .. g_nln_b. */
#define TC_ROWF_BIAS 1
#define TC_ROWF_DROP_PRE 2
#define TC_ROWF_LN_POST 4
#define TC_ROWF_DROP_POST 8
#define TC_ROWF_FILM 16
#define TC_ROWF_RES 32
#define TC_ROWF_STORE_X 64
#define TC_ROWF_NEXT_LN 128
#define TC_ROWF_STORE_H 256    /* hout = T(hn), or T(xn) without NEXT_LN */
#define TC_ROWF_STORE_ROT 512
typedef struct {
    int flags, M, L;           /* rows; rows per sequence (FiLM row = m / L); M % L == 0 */
    const float* z;            /* fp32 [M][512] */
    const float* bias;
    const float* ln_g; const float* ln_b; float ln_eps;
    const float* film; int film_ld;   /* film[seq * film_ld + c] scale, + 512 shift */
    const float* xres;
    float* xout;
    const float* nln_g; const float* nln_b; float nln_eps;
    void* hout; void* rout;
    const float* rope; int pos_mod, pos_base;   /* rotary position = pos_base + m % pos_mod */
    const int* seed; uint32_t drop_thr; float drop_scale; int site_pre, site_post;
    /* backward only */
    const float* d_xn;         /* fp32 [M][512] or NULL: gradient reaching xn through the residual path of the next block */
    const void* d_h;           /* T [M][512] or NULL */
    const void* d_rot;         /* T [M][512] or NULL */
    void* d_z;                 /* T [M][512] */
    float* d_xres;             /* fp32 [M][512] (TC_ROWF_RES) */
    float* d_film; int dfilm_ld;
    float* partials; int chunks;   /* grid = chunks x (M / L) blocks; partials [chunks * M / L][5][512] */
    int dz_f32;                /* != 0: d_z is fp32 (its consumer is not a GEMM: pooled / memory rows of the conditioning path) */
    /* or, instead of `partials` + tcdiff_row_param_reduce: each block adds its sums straight into the gradients (fp32
     * [512] each, NULL = skip; 512 consecutive atomics per block and vector).  g_bias = column sums of d_z: the bias
     * gradient of the nn.Linear that produced z. */
    float* g_bias; float* g_ln_g; float* g_ln_b; float* g_nln_g; float* g_nln_b;
} tcdiff_row_args;
int tcdiff_row_fwd(int dtype, const tcdiff_row_args* a, hipStream_t stream);
int tcdiff_row_bwd(int dtype, const tcdiff_row_args* a, hipStream_t stream);
/* dst[k][c] += sum_blocks partials[blk][k][c] for the non-NULL dst[k], k < 5 */
int tcdiff_row_param_reduce(const float* partials, int n_blocks, float* d_bias, float* d_ln_g, float* d_ln_b,
                            float* d_nln_g, float* d_nln_b, hipStream_t stream);

/* Train-mode attention: as tcdiff_attention (one K/V per sequence), plus dropout on the softmax weights
 * (model/model.py:98; nn.MultiheadAttention dropout) and lse[seq][H][Lp_q] = log2 sum_k 2^(s log2 e) for the backward. */
/* O_lo (bf16 mode, optional): a second token-major image, bf16(o - float(bf16(o))) of the fp32 output o -- with it the backward's
 * delta_i = sum_d dO_id O_id is taken from O + O_lo (16 bits of O instead of 8).  Round 6 found why it matters: delta enters
 * dS = P (dP - delta), a cancellation, and 8-bit O alone put the decoder's self-attention w_qs / w_ks gradients at twice the error the
 * operand rounding of the step explains (tests/test_train_step_gpu.py, oracle._AttnKernelDelta). */
int tcdiff_attention_train(int dtype, const void* Q, const void* K, const void* V, void* O, void* O_lo, float* lse, int n_seq,
                           int H, int Lq, int Lk, int Lp_q, int Lp_k, int ldo, const int* seed, int site, uint32_t drop_thr,
                           float drop_scale, hipStream_t stream);
/* Backward of the above (P is recomputed from Q, K and lse; the dropout bits are regenerated).  dO: head-major image
 * T[n_seq][H][Lp_q][64] (zero pad rows); O: token-major T as written by the forward.  Outputs are token-major T:
 *   dQ[(seq Lq + q) ld_dq + head 64 + d] (times scale_q: the gradient with respect to the UNSCALED projection),
 *   dK[(seq Lk + k) ld_dkv + head 64 + d], dV likewise.   delta: fp32 workspace [n_seq][H][Lp_q].  O_lo: the forward's second
 * image or NULL (delta from the 8-bit O alone).
 * Two launches (query-major for dQ, key-major for dK / dV): no atomics, bitwise reproducible. */
int tcdiff_attention_bwd(int dtype, const void* Q, const void* K, const void* V, const void* O, const void* O_lo,
                         const void* dO,
                         const float* lse, float* delta, void* dQ, int ld_dq, void* dK, void* dV, int ld_dkv, int n_seq,
                         int H, int Lq, int Lk, int Lp_q, int Lp_k, int ldo, float scale_q, const int* seed, int site,
                         uint32_t drop_thr, float drop_scale, hipStream_t stream);

/* small fp32 helpers of the conditioning path and their adjoints */
/* out[r][c] = a[r * ld_a + c] + b[r * ld_b + c], c < cols */
int tcdiff_add_rows(const float* a, int ld_a, const float* b, int ld_b, float* out, int ld_out, int rows, int cols,
                    hipStream_t stream);
/* out[b] = keep[b] ? x[b] : nul (broadcast), rows of `n` floats  (torch.where(keep_mask, ., null_*), model/model.py:585-589,609-610) */
int tcdiff_select_rows(const float* x, const float* nul, const unsigned char* keep, float* out, int B, long n,
                       hipStream_t stream);
/* dx[b] = keep[b] ? g[b] : 0 (dx may be NULL);  dnul += sum over not-kept b of g[b] (caller-zeroed, may be NULL) */
int tcdiff_select_rows_bwd(const float* g, const unsigned char* keep, float* dx, float* dnul, int B, long n,
                           hipStream_t stream);
/* dx[b][s][c] = g_tok[b][s][c] (NULL = 0) + g_pool[b][c] / S: adjoint of {tokens used as they are, tokens.mean(-2)} */
int tcdiff_pool_bwd(const float* g_tok, const float* g_pool, float* dx, int B, int S, int C, hipStream_t stream);

/* out[k] = coef[k] * mean_b terms[b][k] for the four terms of tcdiff_loss_terms (coef = 0.636, 2.964, 0.646, 10.942),
 * out[4] = their sum (model/diffusion.py:735-741). */
int tcdiff_loss_total(const float* terms, int b, float* out, hipStream_t stream);

/* d total / d model_out and d total / d joints_model of the four loss terms (model/diffusion.py:668-741) for
 * total = 0.636 mean_b recon + 2.964 mean_b vel + 0.646 mean_b fk + 10.942 mean_b foot times `gscale` (the incoming
 * gradient of the scalar).  d_out [b][S*dn][C] receives the reconstruction + velocity part (every element written);
 * d_joints [b*S*dn][24][3] the FK + foot part (every element written). */
int tcdiff_loss_terms_bwd(const float* model_out, const float* x_start, const float* joints_model,
                          const float* joints_target, const float* p2_weight, const long* t, const float* gscale,
                          float* d_out, float* d_joints, int b, int dn, int S, int C, int l1, hipStream_t stream);
/* Reverse of tcdiff_ax_from_6v + tcdiff_smpl_fk for the model branch: d_joints -> += d_out[row][4..6] (root) and
 * += d_out[row][7 + 6 j ..] (6-D rotations); motion rows [n][C]. */
int tcdiff_fk_bwd(const float* motion, const float* d_joints, long n, int C, const int* parents, const float* offsets,
                  float* d_out, hipStream_t stream);

/* library identification */
const char* tcdiff_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TCDIFF_HIP_H */
