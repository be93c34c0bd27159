// Training step of the TCDiff denoiser, gfx950: the bandwidth-bound kernels between the GEMMs and attentions of the
// train-mode forward and of the backward pass.  The reference has no code for the latter: it is torch autograd through
// model/model.py:308-402 (decoder layer), :211-245 (music encoder layer), :548-624 (DanceDecoder.forward) and the loss
// of model/diffusion.py:636-741, run by accelerator.backward at TCDiff.py:232.  Here every adjoint is written out and
// fused the way the forward is:
//   cast_transpose   operand repacking for tcdiff_gemm_tile's dgrad / wgrad calls (+ the bias gradient)
//   act_drop         activation + nn.Dropout, forward and backward
//   row_fwd/row_bwd  {bias, dropout, post-LayerNorm, dropout, FiLM, residual, next LayerNorm, rotary} of one 512-wide
//                    row per wave: the whole row-local glue between two GEMMs, and its exact reverse
//   small adjoints   null-conditioning select, mean-pool, loss terms, SMPL chain (fk_math.h)
// Everything is fp32 arithmetic; HBM-bound by design (16-byte accesses, one pass over each operand).
#include "train_common.h"
#include "fk_math.h"
#include "tcdiff_hip.h"

// =====================================================================================================================
// cast / transpose through a 64 x 64 LDS tile
// =====================================================================================================================
template <class T> DEVINL float ld_elem(const T* p);
template <> DEVINL float ld_elem<float>(const float* p) { return *p; }
template <> DEVINL float ld_elem<uint16_t>(const uint16_t* p) { return bf2f(*p); }

template <class P>
DEVINL void store4_T(typename P::elem_t* p, const float (&v)[4]) {
    if (P::IS_BF16) {
        uint2 pk;
        pk.x = pack_bf2(v[0], v[1]);
        pk.y = pack_bf2(v[2], v[3]);
        *reinterpret_cast<uint2*>(p) = pk;
    } else {
        const f32x4_t pk = {v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4_t*>(p) = pk;
    }
}
template <class S>
DEVINL void load4(const S* p, float (&v)[4]) {      // 4 consecutive source elements, 4-element aligned
    if (sizeof(S) == 4) {
        const f32x4_t q = *reinterpret_cast<const f32x4_t*>(p);
        v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
    } else {
        const uint2 q = *reinterpret_cast<const uint2*>(p);
        v[0] = bf2f((uint16_t)(q.x & 0xffffu)); v[1] = bf2f((uint16_t)(q.x >> 16));
        v[2] = bf2f((uint16_t)(q.y & 0xffffu)); v[3] = bf2f((uint16_t)(q.y >> 16));
    }
}

template <class P, class S>
DEVINL void cast_transpose_tile(float (&tile)[64][65], int bx, int by, const S* __restrict__ src, int rows, int cols, int ld_src,
                                typename P::elem_t* __restrict__ dst, int ld_dst, int cols_pad,
                                typename P::elem_t* __restrict__ dstT, int ld_dstT, int rows_pad, float* __restrict__ colsum,
                                int vec_src, int vec_dst, int vec_dstT) {
    const int tid = threadIdx.x;
    const int r0 = by * 64, c0 = bx * 64;
    const int lr = tid >> 4, lc = (tid & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = lr + 16 * i, gr = r0 + row, gc = c0 + lc;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (gr < rows) {
            const S* sp = src + (long)gr * ld_src + gc;
            if (vec_src && gc + 3 < cols) load4<S>(sp, v);
            else
#pragma unroll
                for (int j = 0; j < 4; ++j) if (gc + j < cols) v[j] = ld_elem<S>(sp + j);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) tile[row][lc + j] = v[j];
        if (dst && gr < rows) {
            typename P::elem_t* dp = dst + (long)gr * ld_dst + gc;
            if (vec_dst && gc + 3 < cols_pad) store4_T<P>(dp, v);
            else
#pragma unroll
                for (int j = 0; j < 4; ++j) if (gc + j < cols_pad) dp[j] = P::from_f32(v[j]);
        }
    }
    __syncthreads();
    if (colsum && tid < 64 && c0 + tid < cols) {
        float s = 0.0f;
#pragma unroll 8
        for (int rr = 0; rr < 64; ++rr) s += tile[rr][tid];
        unsafeAtomicAdd(colsum + c0 + tid, s);
    }
    if (dstT) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tc = lr + 16 * i, gc = c0 + tc, gr = r0 + lc;       // tile column tc -> row gc of dstT
            if (gc >= cols) continue;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = tile[lc + j][tc];
            typename P::elem_t* dp = dstT + (long)gc * ld_dstT + gr;
            if (vec_dstT && gr + 3 < rows_pad) store4_T<P>(dp, v);
            else
#pragma unroll
                for (int j = 0; j < 4; ++j) if (gr + j < rows_pad) dp[j] = P::from_f32(v[j]);
        }
    }
}

template <class P, class S>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const S* __restrict__ src, int rows, int cols, int ld_src,
                                                             typename P::elem_t* __restrict__ dst, int ld_dst, int cols_pad,
                                                             typename P::elem_t* __restrict__ dstT, int ld_dstT,
                                                             int rows_pad, float* __restrict__ colsum, int vec_src,
                                                             int vec_dst, int vec_dstT) {
    __shared__ float tile[64][65];
    cast_transpose_tile<P, S>(tile, blockIdx.x, blockIdx.y, src, rows, cols, ld_src, dst, ld_dst, cols_pad, dstT, ld_dstT,
                              rows_pad, colsum, vec_src, vec_dst, vec_dstT);
}

// The same over a TABLE of matrices in one launch (the weight packs of every nn.Linear after an optimizer step: ~125
// matrices, most of them 512 x 512 -- as separate launches 0.7 ms of 5-us kernels): block -> descriptor by bisection over
// the descriptors' first-tile indices (wave-uniform), then the tile of that matrix.
template <class P>
__global__ __launch_bounds__(256) void cast_transpose_multi_kernel(const tcdiff_ct_desc* __restrict__ descs, int n_desc) {
    __shared__ float tile[64][65];
    int lo = 0, hi = n_desc - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (descs[mid].tile0 <= b) lo = mid; else hi = mid - 1;
    }
    const tcdiff_ct_desc d = descs[lo];
    const int t = b - d.tile0;
    cast_transpose_tile<P, float>(tile, t % d.tiles_x, t / d.tiles_x, reinterpret_cast<const float*>(d.src), d.rows, d.cols,
                                  d.ld_src, reinterpret_cast<typename P::elem_t*>(d.dst), d.ld_dst, d.cols_pad,
                                  reinterpret_cast<typename P::elem_t*>(d.dstT), d.ld_dstT, d.rows_pad, nullptr, d.vec & 1,
                                  (d.vec >> 1) & 1, (d.vec >> 2) & 1);
}

static bool al(const void* p, int a) { return (reinterpret_cast<uintptr_t>(p) & (uintptr_t)(a - 1)) == 0; }

extern "C" int tcdiff_cast_transpose(int dtype, int src_f32, const void* src, int rows, int cols, int ld_src, void* dst,
                                     int ld_dst, int cols_pad, void* dstT, int ld_dstT, int rows_pad, float* colsum,
                                     hipStream_t stream) {
    if (!src || rows <= 0 || cols <= 0 || ld_src < cols || (!dst && !dstT && !colsum)) return TC_ERR_ARG;
    if (dtype != TC_DTYPE_BF16 && dtype != TC_DTYPE_F32) return TC_ERR_ARG;
    if (dst && (cols_pad < cols || ld_dst < cols_pad)) return TC_ERR_ARG;
    if (dstT && (rows_pad < rows || ld_dstT < rows_pad)) return TC_ERR_ARG;
    const int es = dtype == TC_DTYPE_BF16 ? 2 : 4, ss = src_f32 ? 4 : es;
    const int span_c = dst ? cols_pad : cols, span_r = dstT ? rows_pad : rows;
    dim3 grid((span_c + 63) / 64, (span_r + 63) / 64);
    const int vs = al(src, 4 * ss) && ld_src % 4 == 0, vd = dst && al(dst, 4 * es) && ld_dst % 4 == 0;
    const int vt = dstT && al(dstT, 4 * es) && ld_dstT % 4 == 0;
#define TC_CT(POL, ST)                                                                                                 \
    hipLaunchKernelGGL((cast_transpose_kernel<POL, ST>), grid, dim3(256), 0, stream, (const ST*)src, rows, cols, ld_src, \
                       (POL::elem_t*)dst, ld_dst, cols_pad, (POL::elem_t*)dstT, ld_dstT, rows_pad, colsum, vs, vd, vt)
    if (dtype == TC_DTYPE_BF16) {
        if (src_f32) TC_CT(MmaBF16, float); else TC_CT(MmaBF16, uint16_t);
    } else {
        TC_CT(MmaF32, float);
    }
#undef TC_CT
    TC_CHECK_LAUNCH();
    return TC_OK;
}

extern "C" int tcdiff_cast_transpose_multi(int dtype, const tcdiff_ct_desc* descs_dev, int n_desc, int n_tiles,
                                           hipStream_t stream) {
    if (!descs_dev || n_desc <= 0 || n_tiles <= 0) return TC_ERR_ARG;
    if (dtype == TC_DTYPE_BF16)
        hipLaunchKernelGGL(cast_transpose_multi_kernel<MmaBF16>, dim3(n_tiles), dim3(256), 0, stream, descs_dev, n_desc);
    else if (dtype == TC_DTYPE_F32)
        hipLaunchKernelGGL(cast_transpose_multi_kernel<MmaF32>, dim3(n_tiles), dim3(256), 0, stream, descs_dev, n_desc);
    else return TC_ERR_ARG;
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// fills the host-side fields a descriptor derives from the others (tiles_x, n_tiles -> return value, vec); tile0 is the
// caller's running sum.  Returns the number of 64 x 64 tiles of this matrix, or a negative error code.
extern "C" int tcdiff_ct_desc_init(int dtype, tcdiff_ct_desc* d) {
    if (!d || !d->src || d->rows <= 0 || d->cols <= 0 || d->ld_src < d->cols || (!d->dst && !d->dstT)) return TC_ERR_ARG;
    if (dtype != TC_DTYPE_BF16 && dtype != TC_DTYPE_F32) return TC_ERR_ARG;
    if (d->dst && (d->cols_pad < d->cols || d->ld_dst < d->cols_pad)) return TC_ERR_ARG;
    if (d->dstT && (d->rows_pad < d->rows || d->ld_dstT < d->rows_pad)) return TC_ERR_ARG;
    const int es = dtype == TC_DTYPE_BF16 ? 2 : 4;
    const int span_c = d->dst ? d->cols_pad : d->cols, span_r = d->dstT ? d->rows_pad : d->rows;
    d->tiles_x = (span_c + 63) / 64;
    const int vs = al(d->src, 16) && d->ld_src % 4 == 0, vd = d->dst && al(d->dst, 4 * es) && d->ld_dst % 4 == 0;
    const int vt = d->dstT && al(d->dstT, 4 * es) && d->ld_dstT % 4 == 0;
    d->vec = vs | (vd << 1) | (vt << 2);
    return d->tiles_x * ((span_r + 63) / 64);
}

// =====================================================================================================================
// activation + dropout, forward and backward (4 consecutive columns per thread)
// =====================================================================================================================
template <class P, class S, bool BWD>
__global__ __launch_bounds__(256) void act_drop_kernel(const S* __restrict__ a, int ld_a, const typename P::elem_t* __restrict__ dy,
                                                       void* __restrict__ out, int ld_o, int rows, int cols, int act,
                                                       const int* __restrict__ seed, int site, uint32_t thr, float dscale,
                                                       int vec) {
    // forward: out = y (T, [rows][ld_o]);  backward: out = da (S, [rows][ld_a]), dy T [rows][ld_o]
    typedef typename P::elem_t T;
    const int ld_w = BWD ? ld_a : ld_o;                  // leading dimension of the tensor being written
    const int quads = (ld_w + 3) / 4;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * quads) return;
    const int row = (int)(i / quads), c4 = (int)(i % quads) * 4;
    const DropCtx dc = drop_ctx(seed, site, thr, dscale);
    float v[4];
    // whole quads inside the valid columns move as ONE access per tensor (the launcher checks alignment: `vec`); the
    // element-wise path is for ragged widths (438-wide music features) -- as scalar 2-byte accesses this kernel ran at a
    // third of the HBM rate
    const bool full = vec && c4 + 3 < cols;
    float xa[4] = {0.f, 0.f, 0.f, 0.f}, ga[4] = {0.f, 0.f, 0.f, 0.f};
    if (full) {
        load4<S>(a + (long)row * ld_a + c4, xa);
        if (BWD) load4<T>(dy + (long)row * ld_o + c4, ga);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c4 + j < cols) {
                xa[j] = ld_elem<S>(a + (long)row * ld_a + c4 + j);
                if (BWD) ga[j] = P::to_f32(dy[(long)row * ld_o + c4 + j]);
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = c4 + j;
        float r = 0.0f;
        if (c < cols) {
            const bool keep = thr ? drop_keep(dc, (uint32_t)row * (uint32_t)cols + (uint32_t)c) : true;
            if (!BWD) r = keep ? apply_act(xa[j], act) * dscale : 0.0f;
            else r = keep ? ga[j] * dscale * act_grad(xa[j], act) : 0.0f;
        }
        v[j] = r;
    }
    if (BWD) {
        S* op = reinterpret_cast<S*>(out) + (long)row * ld_a + c4;
        if (vec && c4 + 3 < ld_a) {
            if (sizeof(S) == 4) *reinterpret_cast<f32x4_t*>(op) = f32x4_t{v[0], v[1], v[2], v[3]};
            else {
                uint2 pk;
                pk.x = pack_bf2(v[0], v[1]);
                pk.y = pack_bf2(v[2], v[3]);
                *reinterpret_cast<uint2*>(op) = pk;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c4 + j < ld_a) {
                    if (sizeof(S) == 4) reinterpret_cast<float*>(op)[j] = v[j];
                    else reinterpret_cast<uint16_t*>(op)[j] = f2bf(v[j]);
                }
        }
    } else {
        T* op = reinterpret_cast<T*>(out) + (long)row * ld_o + c4;
        if (vec && c4 + 3 < ld_o) store4_T<P>(op, v);
        else
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (c4 + j < ld_o) op[j] = P::from_f32(v[j]);
    }
}

template <bool BWD>
static int launch_act_drop(int dtype, int a_f32, const void* a, int ld_a, const void* dy, void* out, int ld_o, int rows,
                           int cols, int act, const int* seed, int site, uint32_t thr, float dscale, hipStream_t stream) {
    if (!a || !out || rows <= 0 || cols <= 0 || ld_a < cols || ld_o < cols || (BWD && !dy)) return TC_ERR_ARG;
    if (dtype != TC_DTYPE_BF16 && dtype != TC_DTYPE_F32) return TC_ERR_ARG;
    if (act < TC_ACT_NONE || act > TC_ACT_SILU) return TC_ERR_ARG;
    if (!thr) dscale = 1.0f;
    const long n = (long)rows * (((BWD ? ld_a : ld_o) + 3) / 4);
    const int vec = al(a, 16) && al(out, 16) && (!dy || al(dy, 16)) && ld_a % 4 == 0 && ld_o % 4 == 0;
    dim3 grid((unsigned)((n + 255) / 256));
#define TC_AD(POL, ST)                                                                                                   \
    hipLaunchKernelGGL((act_drop_kernel<POL, ST, BWD>), grid, dim3(256), 0, stream, (const ST*)a, ld_a,                     \
                       (const POL::elem_t*)dy, out, ld_o, rows, cols, act, seed, site, thr, dscale, vec)
    if (dtype == TC_DTYPE_BF16) {
        if (a_f32) TC_AD(MmaBF16, float); else TC_AD(MmaBF16, uint16_t);
    } else {
        TC_AD(MmaF32, float);
    }
#undef TC_AD
    TC_CHECK_LAUNCH();
    return TC_OK;
}

extern "C" int tcdiff_act_drop(int dtype, int a_f32, const void* a, int ld_a, void* y, int ld_y, int rows, int cols, int act,
                               const int* seed, int site, uint32_t drop_thr, float drop_scale, hipStream_t stream) {
    return launch_act_drop<false>(dtype, a_f32, a, ld_a, nullptr, y, ld_y, rows, cols, act, seed, site, drop_thr, drop_scale,
                                  stream);
}
extern "C" int tcdiff_act_drop_bwd(int dtype, int a_f32, const void* a, int ld_a, const void* dy, int ld_y, void* da,
                                   int rows, int cols, int act, const int* seed, int site, uint32_t drop_thr,
                                   float drop_scale, hipStream_t stream) {
    return launch_act_drop<true>(dtype, a_f32, a, ld_a, dy, da, ld_y, rows, cols, act, seed, site, drop_thr, drop_scale,
                                 stream);
}

// =====================================================================================================================
// PositionalEncoding in train mode (use_rotary=False): x = dropout(x + pe[position]) on fp32 rows, in place
// (model/utils.py:27-32 at model/model.py:564,580; batch_first: row m of a [B, L, D] tensor is position m % L)
// =====================================================================================================================
__global__ __launch_bounds__(256) void pos_drop_kernel(float* __restrict__ x, const float* __restrict__ pe, int rows, int cols,
                                                       int pos_mod, const int* __restrict__ seed, int site, uint32_t thr,
                                                       float dscale) {
    const int quads = cols >> 2;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * quads) return;
    const int row = (int)(i / quads), c4 = (int)(i % quads) * 4;
    const DropCtx dc = drop_ctx(seed, site, thr, dscale);
    f32x4_t v = *reinterpret_cast<const f32x4_t*>(x + (long)row * cols + c4);
    if (pe) v += *reinterpret_cast<const f32x4_t*>(pe + (long)(row % pos_mod) * cols + c4);
    if (thr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = drop_keep(dc, (uint32_t)row * (uint32_t)cols + (uint32_t)(c4 + j)) ? v[j] * dscale : 0.0f;
    }
    *reinterpret_cast<f32x4_t*>(x + (long)row * cols + c4) = v;
}
extern "C" int tcdiff_pos_drop(float* x, int rows, int cols, const float* pe, int pos_mod, const int* seed, int site,
                               uint32_t drop_thr, float drop_scale, hipStream_t stream) {
    if (!x || rows <= 0 || cols <= 0 || cols % 4 || (pe && pos_mod <= 0) || (drop_thr && !seed)) return TC_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (pe && (reinterpret_cast<uintptr_t>(pe) & 15))) return TC_ERR_ALIGN;
    const long n = (long)rows * (cols / 4);
    hipLaunchKernelGGL(pos_drop_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, pe, rows, cols, pe ? pos_mod : 1,
                       seed, site, drop_thr, drop_scale);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// =====================================================================================================================
// row-local block glue: one wave per 512-wide row; lane l owns columns [4l, 4l+4) and [256+4l, 256+4l+4)
// =====================================================================================================================
struct Row8 { f32x4_t a, b; };      // the lane's eight columns

DEVINL Row8 ld_row_f32(const float* p, int c0, int c1) {
    Row8 r;
    r.a = *reinterpret_cast<const f32x4_t*>(p + c0);
    r.b = *reinterpret_cast<const f32x4_t*>(p + c1);
    return r;
}
DEVINL void st_row_f32(float* p, int c0, int c1, const Row8& r) {
    *reinterpret_cast<f32x4_t*>(p + c0) = r.a;
    *reinterpret_cast<f32x4_t*>(p + c1) = r.b;
}
template <class P>
DEVINL Row8 ld_row_T(const void* base, long row, int c0, int c1) {
    typedef typename P::elem_t T;
    const T* p = reinterpret_cast<const T*>(base) + row * 512;
    Row8 r;
    float va[4], vb[4];
    load4<T>(p + c0, va);
    load4<T>(p + c1, vb);
    r.a = f32x4_t{va[0], va[1], va[2], va[3]};
    r.b = f32x4_t{vb[0], vb[1], vb[2], vb[3]};
    return r;
}
template <class P>
DEVINL void st_row_T(void* base, long row, int c0, int c1, const Row8& r) {
    typedef typename P::elem_t T;
    T* p = reinterpret_cast<T*>(base) + row * 512;
    const float va[4] = {r.a[0], r.a[1], r.a[2], r.a[3]}, vb[4] = {r.b[0], r.b[1], r.b[2], r.b[3]};
    store4_T<P>(p + c0, va);
    store4_T<P>(p + c1, vb);
}
DEVINL float row_sum8(const Row8& v) { return (v.a[0] + v.a[1]) + (v.a[2] + v.a[3]) + (v.b[0] + v.b[1]) + (v.b[2] + v.b[3]); }

// x_hat = (v - mean) * rstd (two-pass statistics like ops.hip::ln_rot); returns rstd
DEVINL float ln_normalize(Row8& v, float eps) {
    const float mean = wave_sum(row_sum8(v)) * (1.0f / 512.0f);
    float ss = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        v.a[t] -= mean;
        v.b[t] -= mean;
        ss += v.a[t] * v.a[t] + v.b[t] * v.b[t];
    }
    const float rstd = rsqrtf(wave_sum(ss) * (1.0f / 512.0f) + eps);
    v.a *= rstd;
    v.b *= rstd;
    return rstd;
}
// dx of y = x_hat * g + b given gy:  rstd * (gh - mean(gh) - x_hat * mean(gh * x_hat)),  gh = gy * g
DEVINL Row8 ln_backward(const Row8& xhat, float rstd, const Row8& gy, const Row8& g) {
    Row8 gh;
    gh.a = gy.a * g.a;
    gh.b = gy.b * g.b;
    float s1 = row_sum8(gh), s2 = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) s2 += gh.a[t] * xhat.a[t] + gh.b[t] * xhat.b[t];
    s1 = wave_sum(s1) * (1.0f / 512.0f);
    s2 = wave_sum(s2) * (1.0f / 512.0f);
    Row8 dx;
    dx.a = (gh.a - s1 - xhat.a * s2) * rstd;
    dx.b = (gh.b - s1 - xhat.b * s2) * rstd;
    return dx;
}
DEVINL void drop_row(const DropCtx& dc, uint32_t x0, int c0, int c1, Row8& v) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        v.a[t] = drop_apply(dc, x0 + (uint32_t)(c0 + t), v.a[t]);
        v.b[t] = drop_apply(dc, x0 + (uint32_t)(c1 + t), v.b[t]);
    }
}
// rotary pairs (2j, 2j+1) live inside one float4: cs = {cos0, sin0, cos1, sin1}
DEVINL f32x4_t rot4(const f32x4_t& u, const f32x4_t& cs) {
    return f32x4_t{u[0] * cs[0] - u[1] * cs[1], u[1] * cs[0] + u[0] * cs[1], u[2] * cs[2] - u[3] * cs[3], u[3] * cs[2] + u[2] * cs[3]};
}
DEVINL f32x4_t rot4_T(const f32x4_t& g, const f32x4_t& cs) {      // adjoint (= rotation by the negative angle)
    return f32x4_t{g[0] * cs[0] + g[1] * cs[1], g[1] * cs[0] - g[0] * cs[1], g[2] * cs[2] + g[3] * cs[3], g[3] * cs[2] - g[2] * cs[3]};
}

// forward of the block up to xn (before the next LayerNorm); also returns what the backward needs
struct RowFwd {
    Row8 uhat;     // normalised input of the post-LayerNorm (LN_POST)
    float rstd;
    Row8 y;        // value the FiLM scale multiplies
    Row8 xn;
};
DEVINL RowFwd row_forward(const tcdiff_row_args& a, int m, int c0, int c1, const DropCtx& dpre, const DropCtx& dpost) {
    const int f = a.flags;
    RowFwd o;
    Row8 v = ld_row_f32(a.z + (long)m * 512, c0, c1);
    if (f & TC_ROWF_BIAS) {
        const Row8 b = ld_row_f32(a.bias, c0, c1);
        v.a += b.a;
        v.b += b.b;
    }
    if ((f & TC_ROWF_DROP_PRE) && a.drop_thr) drop_row(dpre, (uint32_t)m * 512u, c0, c1, v);
    o.rstd = 1.0f;
    if (f & TC_ROWF_LN_POST) {
        o.rstd = ln_normalize(v, a.ln_eps);
        o.uhat = v;
        const Row8 g = ld_row_f32(a.ln_g, c0, c1), b = ld_row_f32(a.ln_b, c0, c1);
        v.a = v.a * g.a + b.a;
        v.b = v.b * g.b + b.b;
    }
    if ((f & TC_ROWF_DROP_POST) && a.drop_thr) drop_row(dpost, (uint32_t)m * 512u, c0, c1, v);
    o.y = v;
    if (f & TC_ROWF_FILM) {
        const float* fp = a.film + (long)(m / a.L) * a.film_ld;
        const Row8 s = ld_row_f32(fp, c0, c1), sh = ld_row_f32(fp + 512, c0, c1);
        v.a = (s.a + 1.0f) * v.a + sh.a;
        v.b = (s.b + 1.0f) * v.b + sh.b;
    }
    if (f & TC_ROWF_RES) {
        const Row8 x = ld_row_f32(a.xres + (long)m * 512, c0, c1);
        v.a = x.a + v.a;
        v.b = x.b + v.b;
    }
    o.xn = v;
    return o;
}

template <class P>
__global__ __launch_bounds__(256) void row_fwd_kernel(tcdiff_row_args a) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= a.M) return;
    const int c0 = 4 * lane, c1 = 256 + 4 * lane, f = a.flags;
    const DropCtx dpre = drop_ctx(a.seed, a.site_pre, a.drop_thr, a.drop_scale);
    const DropCtx dpost = drop_ctx(a.seed, a.site_post, a.drop_thr, a.drop_scale);
    const RowFwd o = row_forward(a, m, c0, c1, dpre, dpost);
    if (f & TC_ROWF_STORE_X) st_row_f32(a.xout + (long)m * 512, c0, c1, o.xn);
    Row8 u = o.xn;
    if (f & TC_ROWF_NEXT_LN) {
        ln_normalize(u, a.nln_eps);
        const Row8 g = ld_row_f32(a.nln_g, c0, c1), b = ld_row_f32(a.nln_b, c0, c1);
        u.a = u.a * g.a + b.a;
        u.b = u.b * g.b + b.b;
    }
    if (f & TC_ROWF_STORE_H) st_row_T<P>(a.hout, m, c0, c1, u);
    if (f & TC_ROWF_STORE_ROT) {
        const int pos = a.pos_base + (a.pos_mod > 0 ? m % a.pos_mod : m);
        const Row8 cs = ld_row_f32(a.rope + (long)pos * 512, c0, c1);
        Row8 y;
        y.a = rot4(u.a, cs.a);
        y.b = rot4(u.b, cs.b);
        st_row_T<P>(a.rout, m, c0, c1, y);
    }
}

// grid = (chunks, M / L): a block works inside ONE sequence, so its FiLM gradient goes to one row of d_film.  Eight waves
// per block: a wave walks its rows serially (each row is a chain of loads and two wave reductions), so the launch needs
// waves, not work per wave -- with 4 waves x 8 blocks per sequence (one wave per SIMD at 32 sequences) it ran at a quarter
// of the HBM rate.  196 VGPRs = two waves per SIMD = one block per CU; capped at 128 it spills 364 bytes per lane.
// (Tried: the seven column sums and three constants in LDS with ds_add_f32 per element -- 120 VGPRs, but LDS float atomics
// retire under one lane per clock: 168 us per launch instead of 37.)
constexpr int ROWB_WAVES = 8;
template <class P>
__global__ __launch_bounds__(64 * ROWB_WAVES) void row_bwd_kernel(tcdiff_row_args a) {
    __shared__ float red[ROWB_WAVES / 2][7][512];       // 56 KB: the upper four waves deposit, the lower four add theirs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = 4 * lane, c1 = 256 + 4 * lane, f = a.flags;
    const int seq = blockIdx.y;
    const int per = (a.L + (int)gridDim.x - 1) / (int)gridDim.x;
    const int lo = seq * a.L + blockIdx.x * per;
    const int hi = min(lo + per, (seq + 1) * a.L);
    const DropCtx dpre = drop_ctx(a.seed, a.site_pre, a.drop_thr, a.drop_scale);
    const DropCtx dpost = drop_ctx(a.seed, a.site_post, a.drop_thr, a.drop_scale);
    Row8 zero;
    zero.a = f32x4_t{0.f, 0.f, 0.f, 0.f};
    zero.b = zero.a;
    Row8 acc_bias = zero, acc_g = zero, acc_b = zero, acc_g2 = zero, acc_b2 = zero, acc_s = zero, acc_sh = zero;
    Row8 g_post = zero, g_next = zero, fs = zero;
    if (f & TC_ROWF_LN_POST) g_post = ld_row_f32(a.ln_g, c0, c1);
    if (f & TC_ROWF_NEXT_LN) g_next = ld_row_f32(a.nln_g, c0, c1);
    if (f & TC_ROWF_FILM) {
        fs = ld_row_f32(a.film + (long)seq * a.film_ld, c0, c1);
        fs.a += 1.0f;
        fs.b += 1.0f;
    }
    for (int m = lo + wave; m < hi; m += ROWB_WAVES) {
        const RowFwd o = row_forward(a, m, c0, c1, dpre, dpost);
        // ---- gradient reaching xn -----------------------------------------------------------------------------------
        Row8 gh = zero;                                    // d / d hn (the next LayerNorm's output, or xn itself without it)
        if (a.d_h) gh = ld_row_T<P>(a.d_h, m, c0, c1);
        if (a.d_rot) {
            const int pos = a.pos_base + (a.pos_mod > 0 ? m % a.pos_mod : m);
            const Row8 cs = ld_row_f32(a.rope + (long)pos * 512, c0, c1);
            const Row8 gr = ld_row_T<P>(a.d_rot, m, c0, c1);
            gh.a += rot4_T(gr.a, cs.a);
            gh.b += rot4_T(gr.b, cs.b);
        }
        Row8 gx = zero;
        if (a.d_xn) gx = ld_row_f32(a.d_xn + (long)m * 512, c0, c1);
        if (f & TC_ROWF_NEXT_LN) {
            Row8 xh = o.xn;
            const float rstd2 = ln_normalize(xh, a.nln_eps);
            acc_g2.a += gh.a * xh.a;
            acc_g2.b += gh.b * xh.b;
            acc_b2.a += gh.a;
            acc_b2.b += gh.b;
            const Row8 dx = ln_backward(xh, rstd2, gh, g_next);
            gx.a += dx.a;
            gx.b += dx.b;
        } else {
            gx.a += gh.a;
            gx.b += gh.b;
        }
        if (f & TC_ROWF_RES) st_row_f32(a.d_xres + (long)m * 512, c0, c1, gx);
        // ---- FiLM ---------------------------------------------------------------------------------------------------
        Row8 gy = gx;
        if (f & TC_ROWF_FILM) {
            acc_s.a += gx.a * o.y.a;
            acc_s.b += gx.b * o.y.b;
            acc_sh.a += gx.a;
            acc_sh.b += gx.b;
            gy.a = gx.a * fs.a;
            gy.b = gx.b * fs.b;
        }
        if ((f & TC_ROWF_DROP_POST) && a.drop_thr) drop_row(dpost, (uint32_t)m * 512u, c0, c1, gy);
        // ---- post-LayerNorm -----------------------------------------------------------------------------------------
        Row8 gu = gy;
        if (f & TC_ROWF_LN_POST) {
            acc_g.a += gy.a * o.uhat.a;
            acc_g.b += gy.b * o.uhat.b;
            acc_b.a += gy.a;
            acc_b.b += gy.b;
            gu = ln_backward(o.uhat, o.rstd, gy, g_post);
        }
        if ((f & TC_ROWF_DROP_PRE) && a.drop_thr) drop_row(dpre, (uint32_t)m * 512u, c0, c1, gu);
        acc_bias.a += gu.a;
        acc_bias.b += gu.b;
        if (a.d_z) {
            if (a.dz_f32) st_row_f32(reinterpret_cast<float*>(a.d_z) + (long)m * 512, c0, c1, gu);
            else st_row_T<P>(a.d_z, m, c0, c1, gu);
        }
    }
    // ---- block sums through LDS: [bias, ln_g, ln_b, nln_g, nln_b] -> one partial row per block (folded by row_param_reduce);
    // [FiLM scale, FiLM shift] -> atomics on this sequence's d_film row, issued as consecutive floats per wave instruction
    // (256-byte segments, the full-rate shape; a lane-strided scatter per wave was 8 M slow atomics per launch)
    const bool film = (f & TC_ROWF_FILM) && a.d_film;
    if (!a.partials && !film && !a.g_bias && !a.g_ln_g && !a.g_ln_b && !a.g_nln_g && !a.g_nln_b) return;
    const Row8* accs[7] = {&acc_bias, &acc_g, &acc_b, &acc_g2, &acc_b2, &acc_s, &acc_sh};
    constexpr int HW = ROWB_WAVES / 2;
    if (wave >= HW) {
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            *reinterpret_cast<f32x4_t*>(&red[wave - HW][k][c0]) = accs[k]->a;
            *reinterpret_cast<f32x4_t*>(&red[wave - HW][k][c1]) = accs[k]->b;
        }
    }
    __syncthreads();
    if (wave < HW) {
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            *reinterpret_cast<f32x4_t*>(&red[wave][k][c0]) += accs[k]->a;
            *reinterpret_cast<f32x4_t*>(&red[wave][k][c1]) += accs[k]->b;
        }
    }
    __syncthreads();
    const float* rf = &red[0][0][0];
    constexpr int WS = 7 * 512;
    if (a.partials) {
        float* out = a.partials + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (5 * 512);
        for (int i = threadIdx.x; i < 5 * 512; i += 64 * ROWB_WAVES)
            out[i] = (rf[i] + rf[WS + i]) + (rf[2 * WS + i] + rf[3 * WS + i]);
    }
    float* const gdst[5] = {a.g_bias, a.g_ln_g, a.g_ln_b, a.g_nln_g, a.g_nln_b};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (!gdst[k]) continue;                           // block-uniform
        const int i = threadIdx.x, j = k * 512 + i;       // 512 threads = the 512 columns
        unsafeAtomicAdd(gdst[k] + i, (rf[j] + rf[WS + j]) + (rf[2 * WS + j] + rf[3 * WS + j]));
    }
    if (film) {
        float* dp = a.d_film + (long)seq * a.dfilm_ld;
        for (int i = threadIdx.x; i < 2 * 512; i += 64 * ROWB_WAVES) {
            const int j = 5 * 512 + i;
            unsafeAtomicAdd(dp + i, (rf[j] + rf[WS + j]) + (rf[2 * WS + j] + rf[3 * WS + j]));
        }
    }
}

static int check_row_args(const tcdiff_row_args* a, bool bwd) {
    if (!a || !a->z || a->M <= 0 || a->L <= 0 || a->M % a->L != 0) return TC_ERR_ARG;
    const int f = a->flags;
    if ((f & TC_ROWF_BIAS) && !a->bias) return TC_ERR_ARG;
    if ((f & TC_ROWF_LN_POST) && (!a->ln_g || !a->ln_b)) return TC_ERR_ARG;
    if ((f & TC_ROWF_FILM) && (!a->film || a->film_ld % 4)) return TC_ERR_ARG;
    if ((f & TC_ROWF_RES) && !a->xres) return TC_ERR_ARG;
    if ((f & TC_ROWF_NEXT_LN) && (!a->nln_g || !a->nln_b)) return TC_ERR_ARG;
    if ((f & TC_ROWF_STORE_ROT) && !a->rope) return TC_ERR_ARG;
    if (!bwd) {
        if ((f & TC_ROWF_STORE_X) && !a->xout) return TC_ERR_ARG;
        if ((f & TC_ROWF_STORE_H) && !a->hout) return TC_ERR_ARG;
        if ((f & TC_ROWF_STORE_ROT) && !a->rout) return TC_ERR_ARG;
    } else {
        if (a->chunks <= 0) return TC_ERR_ARG;
        if ((f & TC_ROWF_RES) && !a->d_xres) return TC_ERR_ARG;
        if (a->d_rot && !a->rope) return TC_ERR_ARG;
        if ((f & TC_ROWF_FILM) && a->d_film && a->dfilm_ld % 4) return TC_ERR_ARG;
    }
    const void* ptrs[] = {a->z, a->bias, a->ln_g, a->ln_b, a->film, a->xres, a->xout, a->nln_g, a->nln_b, a->hout, a->rout,
                          a->rope, a->d_xn, a->d_h, a->d_rot, a->d_z, a->d_xres, a->partials, a->g_bias, a->g_ln_g, a->g_ln_b,
                          a->g_nln_g, a->g_nln_b};
    for (const void* p : ptrs)
        if (p && !al(p, 16)) return TC_ERR_ALIGN;
    return TC_OK;
}

extern "C" int tcdiff_row_fwd(int dtype, const tcdiff_row_args* a, hipStream_t stream) {
    if (dtype != TC_DTYPE_BF16 && dtype != TC_DTYPE_F32) return TC_ERR_ARG;
    const int rc = check_row_args(a, false);
    if (rc != TC_OK) return rc;
    dim3 grid((a->M + 3) / 4);
    if (dtype == TC_DTYPE_BF16) hipLaunchKernelGGL(row_fwd_kernel<MmaBF16>, grid, dim3(256), 0, stream, *a);
    else hipLaunchKernelGGL(row_fwd_kernel<MmaF32>, grid, dim3(256), 0, stream, *a);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

extern "C" int tcdiff_row_bwd(int dtype, const tcdiff_row_args* a, hipStream_t stream) {
    if (dtype != TC_DTYPE_BF16 && dtype != TC_DTYPE_F32) return TC_ERR_ARG;
    const int rc = check_row_args(a, true);
    if (rc != TC_OK) return rc;
    dim3 grid(a->chunks, a->M / a->L);
    if (dtype == TC_DTYPE_BF16) hipLaunchKernelGGL(row_bwd_kernel<MmaBF16>, grid, dim3(64 * ROWB_WAVES), 0, stream, *a);
    else hipLaunchKernelGGL(row_bwd_kernel<MmaF32>, grid, dim3(64 * ROWB_WAVES), 0, stream, *a);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// dst[k][c] += sum over blocks of partials[blk][k][c]: 40 workgroups of 64 columns; four row groups per workgroup walk
// the blocks with independent loads in flight (a single serial walk per column took 59 us for 256 blocks), LDS-combined
struct RowReduceDst { float* p[5]; };
__global__ __launch_bounds__(256) void row_param_reduce_kernel(const float* __restrict__ partials, int n_blocks, RowReduceDst d) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;     // col in 0 .. 2559
    const int k = col >> 9;
    if (!d.p[k]) return;                                   // block-uniform: a block's 64 columns share k
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int b = rg;
    for (; b + 12 < n_blocks; b += 16) {
        s0 += partials[(long)b * 2560 + col];
        s1 += partials[(long)(b + 4) * 2560 + col];
        s2 += partials[(long)(b + 8) * 2560 + col];
        s3 += partials[(long)(b + 12) * 2560 + col];
    }
    for (; b < n_blocks; b += 4) s0 += partials[(long)b * 2560 + col];
    red[rg][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (rg == 0) {
        const int c = threadIdx.x;
        d.p[k][col & 511] += (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}

extern "C" int tcdiff_row_param_reduce(const float* partials, int n_blocks, float* d_bias, float* d_ln_g, float* d_ln_b,
                                       float* d_nln_g, float* d_nln_b, hipStream_t stream) {
    if (!partials || n_blocks <= 0) return TC_ERR_ARG;
    RowReduceDst d = {{d_bias, d_ln_g, d_ln_b, d_nln_g, d_nln_b}};
    hipLaunchKernelGGL(row_param_reduce_kernel, dim3(40), dim3(256), 0, stream, partials, n_blocks, d);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// =====================================================================================================================
// small fp32 helpers of the conditioning path
// =====================================================================================================================
__global__ void add_rows_kernel(const float* __restrict__ a, int ld_a, const float* __restrict__ b, int ld_b,
                                float* __restrict__ out, int ld_out, int rows, int cols) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * cols) return;
    const int r = (int)(i / cols), c = (int)(i % cols);
    out[(long)r * ld_out + c] = a[(long)r * ld_a + c] + b[(long)r * ld_b + c];
}
extern "C" int tcdiff_add_rows(const float* a, int ld_a, const float* b, int ld_b, float* out, int ld_out, int rows,
                               int cols, hipStream_t stream) {
    if (!a || !b || !out || rows <= 0 || cols <= 0 || ld_a < cols || ld_b < cols || ld_out < cols) return TC_ERR_ARG;
    const long n = (long)rows * cols;
    hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, a, ld_a, b, ld_b, out,
                       ld_out, rows, cols);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

__global__ void select_rows_kernel(const float* __restrict__ x, const float* __restrict__ nul,
                                   const unsigned char* __restrict__ keep, float* __restrict__ out, int B, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * n) return;
    const int b = (int)(i / n);
    out[i] = keep[b] ? x[i] : nul[i - (long)b * n];
}
extern "C" int tcdiff_select_rows(const float* x, const float* nul, const unsigned char* keep, float* out, int B, long n,
                                  hipStream_t stream) {
    if (!x || !nul || !keep || !out || B <= 0 || n <= 0) return TC_ERR_ARG;
    const long t = (long)B * n;
    hipLaunchKernelGGL(select_rows_kernel, dim3((unsigned)((t + 255) / 256)), dim3(256), 0, stream, x, nul, keep, out, B, n);
    TC_CHECK_LAUNCH();
    return TC_OK;
}
// one thread per element of the broadcast row: a fixed-order sum over the batch (reproducible)
__global__ void select_rows_bwd_kernel(const float* __restrict__ g, const unsigned char* __restrict__ keep,
                                       float* __restrict__ dx, float* __restrict__ dnul, int B, long n) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float s = 0.0f;
    for (int b = 0; b < B; ++b) {
        const float v = g[(long)b * n + e];
        const bool k = keep[b] != 0;
        if (dx) dx[(long)b * n + e] = k ? v : 0.0f;
        if (!k) s += v;
    }
    if (dnul) dnul[e] += s;
}
extern "C" int tcdiff_select_rows_bwd(const float* g, const unsigned char* keep, float* dx, float* dnul, int B, long n,
                                      hipStream_t stream) {
    if (!g || !keep || (!dx && !dnul) || B <= 0 || n <= 0) return TC_ERR_ARG;
    hipLaunchKernelGGL(select_rows_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g, keep, dx, dnul, B, n);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

__global__ void pool_bwd_kernel(const float* __restrict__ g_tok, const float* __restrict__ g_pool, float* __restrict__ dx,
                                int B, int S, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * S * C) return;
    const int c = (int)(i % C);
    const int b = (int)(i / ((long)S * C));
    dx[i] = (g_tok ? g_tok[i] : 0.0f) + g_pool[(long)b * C + c] / (float)S;
}
extern "C" int tcdiff_pool_bwd(const float* g_tok, const float* g_pool, float* dx, int B, int S, int C,
                               hipStream_t stream) {
    if (!g_pool || !dx || B <= 0 || S <= 0 || C <= 0) return TC_ERR_ARG;
    const long n = (long)B * S * C;
    hipLaunchKernelGGL(pool_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, g_tok, g_pool, dx, B, S, C);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// =====================================================================================================================
// loss terms, backward (model/diffusion.py:668-741 differentiated by hand)
// =====================================================================================================================
DEVINL float dloss(float d, int l1) { return l1 ? (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) : 2.0f * d; }

__global__ __launch_bounds__(256) void loss_bwd_out_kernel(const float* __restrict__ mo, const float* __restrict__ xs,
                                                           const float* __restrict__ w, const long* __restrict__ t,
                                                           const float* __restrict__ gscale, float* __restrict__ d_out,
                                                           int b, int dn, int S, int C, int l1) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long Lq = (long)S * dn;
    if (i >= (long)b * Lq * C) return;
    const int c = (int)(i % C);
    const long row = i / C;
    const int bi = (int)(row / Lq);
    const long sd = row - (long)bi * Lq;
    const int d = (int)(sd % dn), s = (int)(sd / dn);
    auto model = [&](int ss) { return mo[((long)bi * Lq + (long)ss * dn + d) * C + c]; };
    auto target = [&](int ss) { return xs[(((long)bi * dn + d) * S + ss) * C + c]; };
    const float gs = gscale ? gscale[0] : 1.0f;
    const float wb = w[t[bi]] / (float)b;
    float g = 0.636f * wb / (float)(Lq * C) * dloss(model(s) - target(s), l1);
    if (c >= 4) {
        const float kv = 2.964f * wb / (float)((long)(S - 1) * dn * (C - 4));
        const float ms = model(s), ts = target(s);
        if (s > 0) g += kv * dloss((ms - model(s - 1)) - (ts - target(s - 1)), l1);
        if (s + 1 < S) g -= kv * dloss((model(s + 1) - ms) - (target(s + 1) - ts), l1);
    }
    d_out[i] = gs * g;
}

__global__ __launch_bounds__(256) void loss_bwd_joints_kernel(const float* __restrict__ mo, const float* __restrict__ jm,
                                                              const float* __restrict__ jt, const float* __restrict__ w,
                                                              const long* __restrict__ t, const float* __restrict__ gscale,
                                                              float* __restrict__ d_j, int b, int dn, int S, int C, int l1) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // (row, joint, k)
    const long Lq = (long)S * dn;
    if (i >= (long)b * Lq * 72) return;
    const int k = (int)(i % 3), j = (int)((i / 3) % 24);
    const long row = i / 72;
    const int bi = (int)(row / Lq);
    const long sd = row - (long)bi * Lq;
    const int d = (int)(sd % dn), s = (int)(sd / dn);
    const float gs = gscale ? gscale[0] : 1.0f;
    const float cf = 0.646f * (w[t[bi]] / (float)b) / (float)(Lq * 69);
    const float cfo = 10.942f / (float)b / (float)(Lq * 12);
    auto JM = [&](long r, int jj) { return jm[(r * 24 + jj) * 3 + k]; };
    auto JT = [&](long r, int jj) { return jt[(r * 24 + jj) * 3 + k]; };
    float g = 0.0f;
    // FK term: mean over joints 1..23 of loss((jm_j - jm_0) - (jt_j - jt_0))
    const float m0 = JM(row, 0), t0 = JT(row, 0);
    if (j > 0) {
        g += cf * dloss((JM(row, j) - m0) - (JT(row, j) - t0), l1);
    } else {
        float s0 = 0.0f;
        for (int jj = 1; jj < 24; ++jj) s0 += dloss((JM(row, jj) - m0) - (JT(row, jj) - t0), l1);
        g -= cf * s0;
    }
    // foot-skate term: v(s) = [contact(s) > 0.95] (foot(s + 1) - foot(s)), s < S - 1
    const int fi = j == 7 ? 0 : (j == 8 ? 1 : (j == 10 ? 2 : (j == 11 ? 3 : -1)));
    if (fi >= 0) {
        if (s > 0 && mo[(row - dn) * C + fi] > 0.95f) g += cfo * dloss(JM(row, j) - JM(row - dn, j), l1);
        if (s + 1 < S && mo[row * C + fi] > 0.95f) g -= cfo * dloss(JM(row + dn, j) - JM(row, j), l1);
    }
    d_j[i] = gs * g;
}

extern "C" int tcdiff_loss_terms_bwd(const float* model_out, const float* x_start, const float* joints_model,
                                     const float* joints_target, const float* p2_weight, const long* t,
                                     const float* gscale, float* d_out, float* d_joints, int b, int dn, int S, int C,
                                     int l1, hipStream_t stream) {
    if (!model_out || !x_start || !joints_model || !joints_target || !p2_weight || !t || !d_out || !d_joints || b <= 0 ||
        dn <= 0 || S < 2 || C <= 7)
        return TC_ERR_ARG;
    const long n1 = (long)b * S * dn * C, n2 = (long)b * S * dn * 72;
    hipLaunchKernelGGL(loss_bwd_out_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, stream, model_out, x_start,
                       p2_weight, t, gscale, d_out, b, dn, S, C, l1);
    hipLaunchKernelGGL(loss_bwd_joints_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, stream, model_out,
                       joints_model, joints_target, p2_weight, t, gscale, d_joints, b, dn, S, C, l1);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// out[k] = coef[k] * mean_b terms[b][k] (k < 4), out[4] = their sum: the `losses` tuple and `sum(losses)` of
// model/diffusion.py:735-741 (a fixed-order sum over the batch by one thread per term)
__global__ void loss_total_kernel(const float* __restrict__ terms, int b, float* __restrict__ out) {
    __shared__ float v[4];
    const int k = threadIdx.x;
    if (k < 4) {
        const float coef[4] = {0.636f, 2.964f, 0.646f, 10.942f};
        float s = 0.0f;
        for (int i = 0; i < b; ++i) s += terms[i * 4 + k];
        v[k] = coef[k] * (s / (float)b);
        out[k] = v[k];
    }
    __syncthreads();
    if (k == 0) out[4] = ((v[0] + v[1]) + v[2]) + v[3];
}
extern "C" int tcdiff_loss_total(const float* terms, int b, float* out, hipStream_t stream) {
    if (!terms || !out || b <= 0) return TC_ERR_ARG;
    hipLaunchKernelGGL(loss_total_kernel, dim3(1), dim3(64), 0, stream, terms, b, out);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// reverse of ax_from_6v + SMPLSkeleton.forward for one pose per thread (fk_math.h); the thread owns its output row
__global__ __launch_bounds__(64) void fk_bwd_kernel(const float* __restrict__ motion, const float* __restrict__ d_j, long n,
                                                    int C, FkSkel sk, float* __restrict__ d_out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* row = motion + i * C;
    float aa[72], g_aa[72], g_root[3];
    for (int j = 0; j < TC_FK_J; ++j) {
        const V3 a = axis_angle_from_quat(quat_from_6d(row + 7 + 6 * j));
        aa[3 * j] = a.x; aa[3 * j + 1] = a.y; aa[3 * j + 2] = a.z;
    }
    fk_backward(aa, sk, d_j + i * 72, g_aa, g_root);
    float* o = d_out + i * C;
    o[4] += g_root[0]; o[5] += g_root[1]; o[6] += g_root[2];
    for (int j = 0; j < TC_FK_J; ++j) {
        float g6[6];
        ax_from_6v_bwd(row + 7 + 6 * j, v3(g_aa[3 * j], g_aa[3 * j + 1], g_aa[3 * j + 2]), g6);
#pragma unroll
        for (int q = 0; q < 6; ++q) o[7 + 6 * j + q] += g6[q];
    }
}

extern "C" int tcdiff_fk_bwd(const float* motion, const float* d_joints, long n, int C, const int* parents,
                             const float* offsets, float* d_out, hipStream_t stream) {
    if (!motion || !d_joints || !parents || !offsets || !d_out || n <= 0 || C < 7 + 6 * TC_FK_J) return TC_ERR_ARG;
    FkSkel sk;
    for (int j = 0; j < TC_FK_J; ++j) sk.has_children[j] = 0;
    for (int j = 0; j < TC_FK_J; ++j) {
        sk.parent[j] = parents[j];
        if (parents[j] >= j) return TC_ERR_ARG;
        if (parents[j] >= 0) sk.has_children[parents[j]] = 1;
        for (int k = 0; k < 3; ++k) sk.off[j][k] = offsets[3 * j + k];
    }
    hipLaunchKernelGGL(fk_bwd_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, motion, d_joints, n, C, sk, d_out);
    TC_CHECK_LAUNCH();
    return TC_OK;
}
