// Row-block GEMMs of the training step (bf16), gfx950: C[M, N] = A[M, K] * Wn[N, K]^T with the epilogues of gemm_tile.
//
// The training step's nn.Linear calls inside the decoder layers (model/model.py:78-80,103,399-401 and their autograd:
// seven forward products and eight input-gradient products per layer, 28 800 token rows at batch 32) are tall and thin:
// K = 512 or 1024, N = 512 .. 1536.  As 128 x 128 tiles (gemm.hip) a K = 512 product is 8 k-tiles between a prologue and an
// epilogue of about the same length (profiles/r04_gemm_shapes.log: 420-590 TFLOP/s).  The sampler's chain kernel runs the
// same products at 2-3 x that rate per CU (one 512 x 512 product over a 64-row block: ~5 us) because the activation block
// is loaded once and only the weights move, wave-private, L2 -> registers.  This kernel is that GEMM phase on its own:
//
//   * a workgroup (8 waves) keeps a block of 16 MT rows of A (MT = 4, 2, 1: as tcdiff_chain) in LDS for all of N; wave w
//     owns columns [512 p + 64 w, +64) of every 512-column phase p (= head w of a Q / K / V / dO image);
//   * the weights arrive as per-wave fragment streams (chain_core.h WStream: 4-KB stages, ring of CH_D in registers), packed
//     ON THE DEVICE once per optimizer step from the fp32 master parameters (pack_row_streams_kernel: forward order from
//     W[N, K], input-gradient order from the same W read transposed);
//   * A2 / split_n: phases whose columns start at or beyond split_n read their rows from a second block (Q, K from
//     rot(norm1 x), V from norm1 x: model/model.py:374-383 in one launch);
//   * epilogues (tcdiff_tile_epi, the meanings of gemm_tile): fp32 rows; bf16 rows; head-major scatter; activation +
//     dropout beside the kept pre-activation (linear1 + GELU + nn.Dropout, model/model.py:399-400); the activation's
//     backward on the way out (the input gradient of linear2 through dropout and GELU').  bf16 rows leave through wave-private
//     LDS staging as 16 bytes per lane, 8 lanes per 128-byte row piece (as store_heads).
// No workgroup barrier after the prologue: a wave's phases touch only its own columns and staging area.
#include "chain_core.h"
#include "train_common.h"

struct RowsArgs {
    const char* A;
    const char* A2;
    const char* wstream;
    int split_n, n_stages, M, N, lda_bytes;
    tcdiff_tile_epi e;
};

DEVINL void unpack8(const u32x4& c, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __builtin_bit_cast(float, c[i] << 16);
        v[2 * i + 1] = __builtin_bit_cast(float, c[i] & 0xFFFF0000u);
    }
}
DEVINL u32x4 pack8(const float (&v)[8]) {
    u32x4 c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = pack_bf2(v[2 * i], v[2 * i + 1]);
    return c;
}

// the wave's [16 MT rows][64 columns] tile as bf16, 32 rows at a time through its 4-KB staging area; sink(m, chunk, value):
// 16 bytes = columns 8 chunk .. of block row m
template <int MT, class F>
DEVINL void staged_rows(const f32x4_t (&acc)[4][MT], char* stg, int lane, F&& sink) {
    lane = fresh_v(lane);
    const int c = lane & 15, g = lane >> 4;
    const int row0 = lane >> 3, ch = lane & 7;
    constexpr int NH = MT == 4 ? 2 : 1, NML = MT == 1 ? 1 : 2;
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
#pragma unroll
        for (int ml = 0; ml < NML; ++ml) {
            const int rl = 16 * ml + c;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4_t v = acc[nt][2 * hf + ml];
                uint2 pk;
                pk.x = pack_bf2(v[0], v[1]);
                pk.y = pack_bf2(v[2], v[3]);
                *reinterpret_cast<uint2*>(stg + rl * 128 + (((2 * nt + (g >> 1)) ^ ((rl >> 1) & 7)) << 4) + 8 * (g & 1)) = pk;
            }
        }
        // (a wave's LDS queue is in order: the reads below see the writes above)
#pragma unroll
        for (int k = 0; k < 2 * NML; ++k) {
            const int row = row0 + 8 * k;
            const u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
            sink(32 * hf + row, ch, v);
        }
    }
}

// KST: 32-deep k-steps of the contraction (16: K = 512, 32: K = 1024); MT: 16-row tiles per block
template <int KST, int MT>
__global__ __launch_bounds__(512) void gemm_rows_kernel(RowsArgs a) {
    constexpr int BR = 16 * MT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * BR;
    const int M = a.M;
    char* abuf = smem + CH_ABUF;
    char* abuf2 = smem + CH_ABUF2;
    const bool two = KST == 16 && a.A2 != nullptr;

    // ---- the block's rows -> LDS (LDS-DMA), the first CH_D weight stages -> registers
    if (wave < 2 * MT) {
#pragma unroll
        for (int kt = 0; kt < KST / 2; ++kt)
            stage_glds<BR, 2 * MT>(abuf + kt * 8192, a.A + kt * TC_ROWB, a.lda_bytes, m0, M, 0, wave, lane);
        if (two) {
#pragma unroll
            for (int kt = 0; kt < 8; ++kt)
                stage_glds<BR, 2 * MT>(abuf2 + kt * 8192, a.A2 + kt * TC_ROWB, a.lda_bytes, m0, M, 0, wave, lane);
        }
    }
    WStream ws;
    ws.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a.wstream) + (long)wave * a.n_stages * CH_STAGE, 0,
                                                a.n_stages * CH_STAGE, 0x00020000);
    ws.voff = (unsigned)lane * 16u;
    ws.pos = 0;
    ws.last = (unsigned)a.n_stages - 1;
#pragma unroll
    for (int i = 0; i < CH_D; ++i) ws_load(ws, i, (unsigned)i);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), visible to the compiler's bookkeeping (chain.hip)
    __syncthreads();

    const tcdiff_tile_epi& e = a.e;
    const int np = a.N >> 9;
    char* stg = stage_area(smem, wave);
    const DropCtx dc = drop_ctx(e.drop_seed, e.drop_site, e.drop_thr, e.drop_scale);
    const float dsc = e.drop_thr ? e.drop_scale : 1.0f;
#pragma unroll 1
    for (int p = 0; p < np; ++p) {
        f32x4_t acc[4][MT];
        zero(acc);
        phase_n512<KST, false, MT>(acc, (two && 512 * p >= a.split_n) ? abuf2 : abuf, ws, lane);
        const int col0 = 512 * p + 64 * wave;            // the wave's first output column of this phase
        const int ln = fresh_v(lane);
        const int c = ln & 15, g = ln >> 4;
        if (e.bias) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4_t b4 = ld4(e.bias + col0 + 16 * nt + 4 * g);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc[nt][mt] += b4;
            }
        }
        if (e.mode == TC_EPI_STORE_F32) {
            // a lane's 16 bytes: columns col0 + 16 nt + 4 g .. of row 16 mt + c; the four lane groups write 64 contiguous bytes
            float* out = reinterpret_cast<float*>(e.out);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = m0 + 16 * mt + c;
                if (m < M) {
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        *reinterpret_cast<f32x4_t*>(out + (long)m * e.ldc + col0 + 16 * nt + 4 * g) = acc[nt][mt];
                }
            }
        } else if (e.mode == TC_EPI_QKV_HEADS) {
            // phase -> image: columns [0, n_q) Q (scaled), [n_q, n_q + n_k) K, the rest V; the wave is the head
            const int n = 512 * p;
            if (n < e.n_q)
                store_heads<true, MT>(acc, e.out, e.scale_q, e.L, e.Lp, e.H, m0, M, wave, lane, smem);
            else if (n < e.n_q + e.n_k)
                store_heads<false, MT>(acc, e.out_k, 1.0f, e.L, e.Lp, e.H, m0, M, wave, lane, smem);
            else
                store_heads<false, MT>(acc, e.out_v, 1.0f, e.L, e.Lp, e.H, m0, M, wave, lane, smem);
        } else if (e.out2) {
            // out = a = T(acc + bias), out2 = T(dropout(act2(a))): nn.Linear + activation + nn.Dropout (model/model.py:399-400)
            uint16_t* o1 = reinterpret_cast<uint16_t*>(e.out);
            uint16_t* o2 = reinterpret_cast<uint16_t*>(e.out2);
            staged_rows<MT>(acc, stg, lane, [&](int row, int ch, const u32x4& val) {
                const int m = m0 + row, n = col0 + 8 * ch;
                if (m >= M) return;
                *reinterpret_cast<u32x4*>(o1 + (long)m * e.ldc + n) = val;
                float x[8], f[8];
                unpack8(val, x);
                if (e.act2 == ACT_GELU) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) f[t] = gelu_erf(x[t]);
                } else if (e.act2 == ACT_RELU) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) f[t] = fmaxf(x[t], 0.0f);
                } else {
#pragma unroll
                    for (int t = 0; t < 8; ++t) f[t] = apply_act(x[t], e.act2);
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) f[t] *= dsc;
                if (e.drop_thr) {
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        f[t] = drop_keep(dc, (uint32_t)m * (uint32_t)a.N + (uint32_t)(n + t)) ? f[t] : 0.0f;
                }
                *reinterpret_cast<u32x4*>(o2 + (long)m * e.ldc2 + n) = pack8(f);
            });
        } else if (e.act_src) {
            // the tile is dY of the activation's output: out = T(T(acc) * mask / (1 - p) * act2'(a)), a from act_src
            uint16_t* o1 = reinterpret_cast<uint16_t*>(e.out);
            const uint16_t* src = reinterpret_cast<const uint16_t*>(e.act_src);
            staged_rows<MT>(acc, stg, lane, [&](int row, int ch, const u32x4& val) {
                const int m = m0 + row, n = col0 + 8 * ch;
                if (m >= M) return;
                const u32x4 sv = *reinterpret_cast<const u32x4*>(src + (long)m * e.ld_src + n);
                float x[8], y[8], f[8];
                unpack8(sv, x);
                unpack8(val, y);
#pragma unroll
                for (int t = 0; t < 8; ++t) y[t] *= dsc;
                if (e.act2 == ACT_GELU) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) f[t] = y[t] * gelu_grad(x[t]);
                } else if (e.act2 == ACT_RELU) {
#pragma unroll
                    for (int t = 0; t < 8; ++t) f[t] = x[t] > 0.0f ? y[t] : 0.0f;
                } else {
#pragma unroll
                    for (int t = 0; t < 8; ++t) f[t] = y[t] * act_grad(x[t], e.act2);
                }
                if (e.drop_thr) {
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        f[t] = drop_keep(dc, (uint32_t)m * (uint32_t)a.N + (uint32_t)(n + t)) ? f[t] : 0.0f;
                }
                *reinterpret_cast<u32x4*>(o1 + (long)m * e.ldc + n) = pack8(f);
            });
        } else {
            uint16_t* o1 = reinterpret_cast<uint16_t*>(e.out);
            staged_rows<MT>(acc, stg, lane, [&](int row, int ch, const u32x4& val) {
                const int m = m0 + row;
                if (m < M) *reinterpret_cast<u32x4*>(o1 + (long)m * e.ldc + col0 + 8 * ch) = val;
            });
        }
    }
}

// =================================================================================================
// weight streams from the fp32 master parameters
// =================================================================================================
// One 16-byte chunk of a stream per thread.  Stream layout ([8 waves][N / 512 phases x K / 32 stages][4096 B], chain.hip /
// engine.py _stages_n512): chunk (w, p, ks, nt, g, c) = the 8 values Wn[512 p + 64 w + 16 nt + c][32 ks + 8 PI(g) + j], j = 0 .. 7,
// with Wn[n][k] = src[n sn + k sk] -- (sn, sk) = (ld, 1) packs W for the forward product, (1, ld) packs the same W for the
// input gradient (output columns = in-features, contraction over out-features).  A table entry may be a PIECE of the packed
// matrix (one of the stacked w_qs / w_ks / w_vs: a phase range going forward, a k-step range in the input gradient): it lands at
// phase p0 + p, k-step ks0 + ks of a stream with np_dst phases of kst_dst stages.
__global__ __launch_bounds__(256) void pack_row_streams_kernel(const tcdiff_ws_desc* __restrict__ descs) {
    const tcdiff_ws_desc d = descs[blockIdx.y];
    const unsigned o = blockIdx.x * 256u + threadIdx.x;
    if (o >= (unsigned)d.N * (unsigned)(d.K >> 3)) return;      // (block-uniform: N K / 8 is a multiple of 256)
    const unsigned kst = (unsigned)d.K >> 5, np = (unsigned)d.N >> 9;
    const int c = o & 15, g = (o >> 4) & 3, nt = (o >> 6) & 3;
    const unsigned q = o >> 8;
    const unsigned ks = q % kst, p = (q / kst) % np, w = q / (kst * np);
    const int pg = (0x9C >> (2 * g)) & 3;
    const long n = 512 * (int)p + 64 * (int)w + 16 * nt + c, k0 = 32 * (int)ks + 8 * pg;
    const float* src = d.src + n * d.sn + k0 * d.sk;
    float v[8];
    if (d.sk == 1) {
        const f32x4_t a = ld4(src), b = ld4(src + 4);
        v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[j * d.sk];
    }
    const unsigned long od = ((((unsigned long)w * d.np_dst + d.p0 + p) * d.kst_dst + d.ks0 + ks) << 8) + (o & 255u);
    reinterpret_cast<u32x4*>(d.dst)[od] = pack8(v);
}

// =================================================================================================
// C ABI
// =================================================================================================
static bool rows_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int tcdiff_gemm_rows(const void* A, const void* A2, int split_n, const void* wstream, int M, int N, int K, int lda,
                                const tcdiff_tile_epi* epi, int mt, hipStream_t stream) {
    if (!A || !wstream || !epi || M <= 0) return TC_ERR_ARG;
    if ((K != 512 && K != 1024) || N <= 0 || N % 512) return TC_ERR_UNSUPPORTED;
    if (lda < K || lda % 8 || !rows_al16(A) || !rows_al16(wstream) || (A2 && !rows_al16(A2))) return TC_ERR_ALIGN;
    if ((long)M * lda * 2 >= (1L << 32)) return TC_ERR_ARG;             // rows are staged with 32-bit offsets
    if (A2 && (K != 512 || split_n <= 0 || split_n % 512)) return TC_ERR_ARG;
    const tcdiff_tile_epi& e = *epi;
    if (e.act != TC_ACT_NONE || e.hgroup != 0) return TC_ERR_UNSUPPORTED;
    if (e.bias && !rows_al16(e.bias)) return TC_ERR_ALIGN;
    if (e.mode == TC_EPI_STORE_F32) {
        if (!e.out || e.ldc < N || e.ldc % 4 || !rows_al16(e.out) || e.out2 || e.act_src) return TC_ERR_ARG;
    } else if (e.mode == TC_EPI_STORE_T) {
        if (!e.out || e.ldc < N || e.ldc % 8 || !rows_al16(e.out)) return TC_ERR_ARG;
        if (e.out2 && (e.act_src || e.ldc2 < N || e.ldc2 % 8 || !rows_al16(e.out2))) return TC_ERR_ARG;
        if (e.act_src && (e.ld_src < N || e.ld_src % 8 || !rows_al16(e.act_src))) return TC_ERR_ARG;
    } else if (e.mode == TC_EPI_QKV_HEADS) {
        // one 512-column phase per image (the wave is the head)
        if (e.H != 8 || e.L < 64 || e.Lp < e.L || (e.n_q != 0 && e.n_q != 512) || (e.n_k != 0 && e.n_k != 512) ||
            N - e.n_q - e.n_k > 512 || e.tok_off || e.seq_off)
            return TC_ERR_ARG;
        if ((e.n_q > 0 && !e.out) || (e.n_k > 0 && !e.out_k) || (N > e.n_q + e.n_k && !e.out_v)) return TC_ERR_ARG;
    } else {
        return TC_ERR_UNSUPPORTED;
    }
    static tc_dev_state dev_state;
    const int n_cu = tc_device_once(dev_state, [](int) {
        const void* fns[6] = {reinterpret_cast<const void*>(gemm_rows_kernel<16, 4>), reinterpret_cast<const void*>(gemm_rows_kernel<16, 2>),
                              reinterpret_cast<const void*>(gemm_rows_kernel<16, 1>), reinterpret_cast<const void*>(gemm_rows_kernel<32, 4>),
                              reinterpret_cast<const void*>(gemm_rows_kernel<32, 2>), reinterpret_cast<const void*>(gemm_rows_kernel<32, 1>)};
        for (const void* f : fns) {
            hipError_t er = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, CH_SMEM);
            if (er != hipSuccess) return er;
        }
        return hipSuccess;
    });
    if (n_cu < 0) return n_cu;
    if (mt == 0) mt = (M + 15) / 16 <= n_cu ? 1 : (M + 31) / 32 <= n_cu ? 2 : 4;
    if (mt != 1 && mt != 2 && mt != 4) return TC_ERR_ARG;
    RowsArgs a;
    a.A = reinterpret_cast<const char*>(A);
    a.A2 = reinterpret_cast<const char*>(A2);
    a.wstream = reinterpret_cast<const char*>(wstream);
    a.split_n = split_n;
    a.n_stages = (N / 512) * (K / 32);
    a.M = M;
    a.N = N;
    a.lda_bytes = lda * 2;
    a.e = e;
    const dim3 grid((M + 16 * mt - 1) / (16 * mt));
#define ROWS_LAUNCH(KST_)                                                                                             \
    do {                                                                                                              \
        if (mt == 4) hipLaunchKernelGGL((gemm_rows_kernel<KST_, 4>), grid, dim3(512), CH_SMEM, stream, a);            \
        else if (mt == 2) hipLaunchKernelGGL((gemm_rows_kernel<KST_, 2>), grid, dim3(512), CH_SMEM, stream, a);       \
        else hipLaunchKernelGGL((gemm_rows_kernel<KST_, 1>), grid, dim3(512), CH_SMEM, stream, a);                    \
    } while (0)
    if (K == 512) ROWS_LAUNCH(16);
    else ROWS_LAUNCH(32);
#undef ROWS_LAUNCH
    TC_CHECK_LAUNCH();
    return TC_OK;
}

extern "C" int tcdiff_pack_row_streams(const tcdiff_ws_desc* descs_dev, int n_desc, int max_elems, hipStream_t stream) {
    // max_elems: the largest N * K of the table (every entry: N % 512 == 0, K % 32 == 0, checked by the caller who filled it)
    if (!descs_dev || n_desc <= 0 || n_desc > 65535 || max_elems <= 0 || max_elems % 2048) return TC_ERR_ARG;
    hipLaunchKernelGGL(pack_row_streams_kernel, dim3((unsigned)(max_elems / 2048), (unsigned)n_desc), dim3(256), 0, stream, descs_dev);
    TC_CHECK_LAUNCH();
    return TC_OK;
}
