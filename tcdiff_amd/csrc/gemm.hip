// MFMA GEMMs of the TCDiff denoiser: C[M,N] = A[M,K] * W[N,K]^T (+ fused epilogues).
//
// A is an activation matrix (row-major, K contiguous), W is an nn.Linear weight as stored by torch
// ([out, in] = [N, K], K contiguous), so both operands stage as rows of K and every MFMA fragment is a
// single 16-byte LDS read (common.h).  Two kernels:
//
//   gemm_tile   128x128 block tile, 4 waves (2x2), general epilogues (bias/act/store, head-major QKV
//               scatter for the attention kernel).  Replaces the aten addmm/linear calls of
//               model/model.py:78-80 (w_qs/w_ks/w_vs), :399 (linear1), :522-528 (fusion projection),
//               :560 (input_projection), :623 (final_layer), :454-465 (time MLP), :164-166 (FiLM).
//   gemm_rowln  64x512 block tile, 8 waves (2x4): one workgroup owns complete 512-wide rows, so the
//               LayerNorm / FiLM / residual that follow fc, linear2 and linear3 in the reference
//               (model/model.py:103-106,327,334,339,344) run in the epilogue from the accumulators,
//               and the NEXT op's LayerNorm(+rotary) input is emitted too (model/model.py:326,332,338,
//               375,387).  No intermediate ever round-trips HBM between the GEMM and its norm.
#include "common.h"
#include "tcdiff_hip.h"

// =================================================================================================
// staging: global -> registers -> LDS (swizzled), 16 B per thread per access
// =================================================================================================
// Per-thread staging registers are plain local arrays indexed only by unrolled constants.
#define STG_PER(ROWS, NT) (((ROWS) * 8 + (NT) - 1) / (NT))

// src: base pointer of row 0 / k-chunk 0 for this k-tile; ld_bytes: row stride in bytes.
// rows >= row_limit are clamped to row_limit-1 (their results are never stored).
template <int ROWS, int NT>
DEVINL void stage_load(u32x4 (&r)[STG_PER(ROWS, NT)], const char* src, long ld_bytes, int row0, int row_limit,
                       int row_mod, int tid) {
#pragma unroll
    for (int i = 0; i < STG_PER(ROWS, NT); ++i) {
        int c = tid + i * NT;
        int row = c >> 3, ch = c & 7;
        int gr = row0 + row;
        gr = gr < row_limit ? gr : row_limit - 1;
        if (row_mod > 0) gr = gr % row_mod;
        r[i] = *reinterpret_cast<const u32x4*>(src + (long)gr * ld_bytes + ch * 16);
    }
}
template <int ROWS, int NT>
DEVINL void stage_store(const u32x4 (&r)[STG_PER(ROWS, NT)], char* lds, int tid) {
    static_assert((ROWS * 8) % NT == 0, "tile chunks must divide evenly over the threads");
#pragma unroll
    for (int i = 0; i < STG_PER(ROWS, NT); ++i) {
        int c = tid + i * NT;
        *reinterpret_cast<u32x4*>(lds + tile_off(c >> 3, c & 7)) = r[i];
    }
}

DEVINL u32x4 lds_frag(const char* tile, int row, int chunk) {
    return *reinterpret_cast<const u32x4*>(tile + tile_off(row, chunk));
}

// =================================================================================================
// gemm_tile: 128 x 128 x (128 B of K) tiles, 256 threads
// =================================================================================================
// position of key `tok` along the key axis of the V^T image (see attention.hip): bf16 swaps bits 2
// and 3 of the index inside each group of 16 keys so that a P^T accumulator tile feeds the PV MFMA
// as its B operand with no lane movement; fp32 keeps natural order.
template <class P>
DEVINL int vt_pos(int tok) {
    if (P::IS_BF16) {
        int kk = tok & 15;
        return (tok & ~15) | (((kk >> 2) & 1) << 3) | ((kk >> 3) << 2) | (kk & 3);
    }
    return tok;
}

template <class P>
__global__ __launch_bounds__(256) void gemm_tile_kernel(const char* __restrict__ A, const char* __restrict__ A2,
                                                        int split_n, const char* __restrict__ W, int M, int N,
                                                        int K, long lda_b, long ldw_b, int a_mod, tcdiff_tile_epi e) {
    typedef typename P::elem_t T;
    __shared__ __attribute__((aligned(16))) char smem[2 * 2 * 128 * TC_ROWB];  // 64 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
    const char* Ause = (A2 != nullptr && n0 >= split_n) ? A2 : A;

    u32x4 sa[STG_PER(128, 256)], sw[STG_PER(128, 256)];
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

    const int nk = K / P::KT;
    constexpr int STAGE = 2 * 128 * TC_ROWB;  // [A tile | W tile]
    constexpr int WOFF = 128 * TC_ROWB;

    stage_load<128, 256>(sa, Ause, lda_b, m0, M, a_mod, tid);
    stage_load<128, 256>(sw, W, ldw_b, n0, N, 0, tid);
    stage_store<128, 256>(sa, smem, tid);
    stage_store<128, 256>(sw, smem + WOFF, tid);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            stage_load<128, 256>(sa, Ause + (long)(kt + 1) * TC_ROWB, lda_b, m0, M, a_mod, tid);
            stage_load<128, 256>(sw, W + (long)(kt + 1) * TC_ROWB, ldw_b, n0, N, 0, tid);
        }
        const char* ta = smem + cur * STAGE + (wm * 64) * TC_ROWB;
        const char* tw = smem + cur * STAGE + WOFF + (wn * 64) * TC_ROWB;
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
                for (int j = 0; j < 2; ++j) P::mma(acc[i][j], fa[i], fw[j]);
        }
        if (kt + 1 < nk) {
            stage_store<128, 256>(sa, smem + (cur ^ 1) * STAGE, tid);
            stage_store<128, 256>(sw, smem + (cur ^ 1) * STAGE + WOFF, tid);
        }
        __syncthreads();
    }

    // ---- epilogue --------------------------------------------------------------------------------
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn * 64 + j * 32 + r;
        const bool n_ok = n < N;
        const float bv = (e.bias && n_ok) ? e.bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mb = m0 + wm * 64 + i * 32;
            if (e.mode == TC_EPI_STORE_T) {
                T* out = reinterpret_cast<T*>(e.out);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    int m = mb + acc_row(q, h);
                    if (m < M && n_ok) out[(long)m * e.ldc + n] = P::from_f32(apply_act(acc[i][j][q] + bv, e.act));
                }
            } else if (e.mode == TC_EPI_STORE_F32) {
                float* out = reinterpret_cast<float*>(e.out);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    int m = mb + acc_row(q, h);
                    if (m < M && n_ok) out[(long)m * e.ldc + n] = apply_act(acc[i][j][q] + bv, e.act);
                }
            } else {  // TC_EPI_QKV_HEADS
                // column n -> (which of Q/K/V, head, d); row m -> (sequence, token)
                int which = n < e.n_q ? 0 : (n < e.n_q + e.n_k ? 1 : 2);
                int nn = n - (which == 0 ? 0 : (which == 1 ? e.n_q : e.n_q + e.n_k));
                int head = nn >> 6, d = nn & 63;
                float sc = which == 0 ? e.scale_q : 1.0f;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    int m = mb + acc_row(q, h);
                    if (m >= M || !n_ok) continue;
                    int seq = m / e.L + e.seq_off, tok = m % e.L + e.tok_off;
                    T v = P::from_f32((acc[i][j][q] + bv) * sc);
                    if (which < 2) {
                        T* dst = reinterpret_cast<T*>(which == 0 ? e.out : e.out_k);
                        dst[(((long)seq * e.H + head) * e.Lp + tok) * 64 + d] = v;
                    } else {
                        T* dst = reinterpret_cast<T*>(e.out_vt);
                        dst[(((long)seq * e.H + head) * 64 + d) * e.Lp + vt_pos<P>(tok)] = v;
                    }
                }
            }
        }
    }
}

// =================================================================================================
// gemm_rowln: 64 x 512 tiles, 512 threads; row-complete epilogue
// =================================================================================================
#define ROWLN_SMEM (2 * (64 + 512) * TC_ROWB + 4 * 64 * 4 * 4)

template <class P>
__global__ __launch_bounds__(512) void gemm_rowln_kernel(const char* __restrict__ A, const char* __restrict__ W,
                                                         int M, int K, long lda_b, long ldw_b, int a_mod,
                                                         tcdiff_row_epi e) {
    typedef typename P::elem_t T;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 64;

    constexpr int STAGE = (64 + 512) * TC_ROWB;  // [A tile | W tile]
    constexpr int WOFF = 64 * TC_ROWB;
    float* red = reinterpret_cast<float*>(smem + 2 * STAGE);  // [4][64][4]

    u32x4 sa[STG_PER(64, 512)], sw[STG_PER(512, 512)];
    f32x16_t acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.0f;

    const int nk = K / P::KT;
    stage_load<64, 512>(sa, A, lda_b, m0, M, a_mod, tid);
    stage_load<512, 512>(sw, W, ldw_b, 0, 512, 0, tid);
    stage_store<64, 512>(sa, smem, tid);
    stage_store<512, 512>(sw, smem + WOFF, tid);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            stage_load<64, 512>(sa, A + (long)(kt + 1) * TC_ROWB, lda_b, m0, M, a_mod, tid);
            stage_load<512, 512>(sw, W + (long)(kt + 1) * TC_ROWB, ldw_b, 0, 512, 0, tid);
        }
        const char* ta = smem + cur * STAGE + (wm * 32) * TC_ROWB;
        const char* tw = smem + cur * STAGE + WOFF + (wn * 128) * TC_ROWB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            u32x4 fa = lds_frag(ta, r, 2 * ks + h);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                u32x4 fw = lds_frag(tw, j * 32 + r, 2 * ks + h);
                P::mma(acc[j], fa, fw);
            }
        }
        if (kt + 1 < nk) {
            stage_store<64, 512>(sa, smem + (cur ^ 1) * STAGE, tid);
            stage_store<512, 512>(sw, smem + (cur ^ 1) * STAGE + WOFF, tid);
        }
        __syncthreads();
    }

    // ---- row-complete epilogue ---------------------------------------------------------------------
    // this lane: columns n_j = wn*128 + j*32 + r (j=0..3); rows lr_q = wm*32 + acc_row(q,h) (q=0..15)
    const int f = e.flags;
    int ncol[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ncol[j] = wn * 128 + j * 32 + r;

    if (f & TC_ROW_BIAS) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float b = e.bias[ncol[j]];
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[j][q] += b;
        }
    }

    // full-row (512-wide) sum of a per-lane quantity: in-lane over j, 32-lane shuffle, 4 waves via LDS
    auto row_reduce = [&](float (&part)[16], float* scratch) {
#pragma unroll
        for (int q = 0; q < 16; ++q) part[q] = half_sum(part[q]);
        if (r == 0) {
#pragma unroll
            for (int q = 0; q < 16; ++q) scratch[(wm * 32 + acc_row(q, h)) * 4 + wn] = part[q];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float4 v = *reinterpret_cast<const float4*>(&scratch[(wm * 32 + acc_row(q, h)) * 4]);
            part[q] = (v.x + v.y) + (v.z + v.w);
        }
    };
    auto layer_norm_rows = [&](float eps, const float* g, const float* b, float* scratch0, float* scratch1) {
        float part[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) part[q] = (acc[0][q] + acc[1][q]) + (acc[2][q] + acc[3][q]);
        row_reduce(part, scratch0);
        float mean[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) mean[q] = part[q] * (1.0f / 512.0f);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            float s = 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float d = acc[j][q] - mean[q];
                s += d * d;
            }
            part[q] = s;
        }
        row_reduce(part, scratch1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gg = g[ncol[j]], bb = b[ncol[j]];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float rstd = rsqrtf(part[q] * (1.0f / 512.0f) + eps);
                acc[j][q] = (acc[j][q] - mean[q]) * rstd * gg + bb;
            }
        }
    };

    if (f & TC_ROW_LN_POST) layer_norm_rows(e.ln_eps, e.ln_g, e.ln_b, red, red + 256);

    if (f & (TC_ROW_FILM | TC_ROW_RES)) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            int m = m0 + wm * 32 + acc_row(q, h);
            int mc = m < M ? m : M - 1;
            int seq = mc / e.L;
            int mr = e.xres_mod > 0 ? mc % e.xres_mod : mc;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[j][q];
                if (f & TC_ROW_FILM) {
                    const float* fp = e.film + (long)seq * e.film_ld + ncol[j];
                    v = (fp[0] + 1.0f) * v + fp[512];
                }
                acc[j][q] = e.xres[(long)mr * 512 + ncol[j]] + v;
            }
        }
    }

    if (f & TC_ROW_STORE_X) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            int m = m0 + wm * 32 + acc_row(q, h);
            if (m < M) {
                long mo = (long)m * e.out_mul + e.out_add;
#pragma unroll
                for (int j = 0; j < 4; ++j) e.xout[mo * 512 + ncol[j]] = acc[j][q];
            }
        }
    }

    if (!(f & TC_ROW_NEXT_LN) && (f & TC_ROW_STORE_H)) {  // plain T copy of v (A operand of final_layer)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            int m = m0 + wm * 32 + acc_row(q, h);
            if (m < M) {
                long mo = (long)m * e.out_mul + e.out_add;
#pragma unroll
                for (int j = 0; j < 4; ++j) reinterpret_cast<T*>(e.hout)[mo * 512 + ncol[j]] = P::from_f32(acc[j][q]);
            }
        }
    }

    if (f & TC_ROW_NEXT_LN) {
        layer_norm_rows(e.nln_eps, e.nln_g, e.nln_b, red + 512, red + 768);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            int m = m0 + wm * 32 + acc_row(q, h);
            long mo = (long)m * e.out_mul + e.out_add;
            int pos = (int)(mo % e.L);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = acc[j][q];
                float partner = __shfl_xor(v, 1);  // the other element of the rotary pair (adjacent column)
                if (m < M) {
                    if (f & TC_ROW_STORE_H) reinterpret_cast<T*>(e.hout)[mo * 512 + ncol[j]] = P::from_f32(v);
                    if (f & TC_ROW_STORE_ROT) {
                        int n = ncol[j];
                        const float* cs = e.rope + (long)pos * 512 + (n & ~1);
                        float c = cs[0], s = cs[1];
                        float y = (n & 1) ? (v * c + partner * s) : (v * c - partner * s);
                        reinterpret_cast<T*>(e.rout)[mo * 512 + n] = P::from_f32(y);
                    }
                }
            }
        }
    }
}

// =================================================================================================
// C ABI
// =================================================================================================
static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int tcdiff_gemm_tile(int dtype, const void* A, const void* A2, int split_n, const void* W, int M, int N,
                                int K, int lda, int ldw, int a_mod, const tcdiff_tile_epi* epi,
                                hipStream_t stream) {
    if (!A || !W || !epi || M <= 0 || N <= 0 || K <= 0) return TC_ERR_ARG;
    const int es = dtype == TC_DTYPE_BF16 ? 2 : 4;
    const int kt = dtype == TC_DTYPE_BF16 ? 64 : 32;
    if (dtype != TC_DTYPE_BF16 && dtype != TC_DTYPE_F32) return TC_ERR_ARG;
    if (K % kt != 0) return TC_ERR_ARG;
    if (!aligned16(A) || !aligned16(W) || (A2 && !aligned16(A2)) || ((long)lda * es) % 16 || ((long)ldw * es) % 16)
        return TC_ERR_ALIGN;
    if (A2 && (split_n % 128 != 0)) return TC_ERR_ARG;
    if (epi->mode == TC_EPI_QKV_HEADS && (epi->L <= 0 || epi->Lp <= 0 || epi->H <= 0)) return TC_ERR_ARG;
    tcdiff_tile_epi e = *epi;
    dim3 grid((N + 127) / 128, (M + 127) / 128);
    if (dtype == TC_DTYPE_BF16)
        hipLaunchKernelGGL(gemm_tile_kernel<MmaBF16>, grid, dim3(256), 0, stream, (const char*)A, (const char*)A2,
                           split_n, (const char*)W, M, N, K, (long)lda * es, (long)ldw * es, a_mod, e);
    else
        hipLaunchKernelGGL(gemm_tile_kernel<MmaF32>, grid, dim3(256), 0, stream, (const char*)A, (const char*)A2,
                           split_n, (const char*)W, M, N, K, (long)lda * es, (long)ldw * es, a_mod, e);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

extern "C" int tcdiff_gemm_rowln(int dtype, const void* A, const void* W, int M, int K, int lda, int ldw, int a_mod,
                                 const tcdiff_row_epi* epi, hipStream_t stream) {
    if (!A || !W || !epi || M <= 0 || K <= 0) return TC_ERR_ARG;
    if (dtype != TC_DTYPE_BF16 && dtype != TC_DTYPE_F32) return TC_ERR_ARG;
    const int es = dtype == TC_DTYPE_BF16 ? 2 : 4;
    const int kt = dtype == TC_DTYPE_BF16 ? 64 : 32;
    if (K % kt != 0) return TC_ERR_ARG;
    if (!aligned16(A) || !aligned16(W) || ((long)lda * es) % 16 || ((long)ldw * es) % 16) return TC_ERR_ALIGN;
    const int f = epi->flags;
    if ((f & TC_ROW_BIAS) && !epi->bias) return TC_ERR_ARG;
    if ((f & TC_ROW_LN_POST) && (!epi->ln_g || !epi->ln_b)) return TC_ERR_ARG;
    if ((f & TC_ROW_FILM) && !epi->film) return TC_ERR_ARG;
    if ((f & (TC_ROW_FILM | TC_ROW_RES)) && !epi->xres) return TC_ERR_ARG;
    if ((f & TC_ROW_STORE_X) && !epi->xout) return TC_ERR_ARG;
    if ((f & TC_ROW_NEXT_LN) && (!epi->nln_g || !epi->nln_b)) return TC_ERR_ARG;
    if ((f & TC_ROW_STORE_H) && !epi->hout) return TC_ERR_ARG;
    if ((f & TC_ROW_STORE_ROT) && (!epi->rout || !epi->rope)) return TC_ERR_ARG;
    if (epi->L <= 0) return TC_ERR_ARG;
    tcdiff_row_epi e = *epi;
    if (e.out_mul <= 0) e.out_mul = 1;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rowln_kernel<MmaBF16>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, ROWLN_SMEM);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rowln_kernel<MmaF32>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, ROWLN_SMEM);
        attr_set = true;
    }
    dim3 grid((M + 63) / 64);
    if (dtype == TC_DTYPE_BF16)
        hipLaunchKernelGGL(gemm_rowln_kernel<MmaBF16>, grid, dim3(512), ROWLN_SMEM, stream, (const char*)A,
                           (const char*)W, M, K, (long)lda * es, (long)ldw * es, a_mod, e);
    else
        hipLaunchKernelGGL(gemm_rowln_kernel<MmaF32>, grid, dim3(512), ROWLN_SMEM, stream, (const char*)A,
                           (const char*)W, M, K, (long)lda * es, (long)ldw * es, a_mod, e);
    TC_CHECK_LAUNCH();
    return TC_OK;
}
