// Bandwidth-bound helpers of the TCDiff hot path (gfx950): LayerNorm(+rotary) prologue, rotary table,
// conditioning-path elementwise ops, per-step K/V scatter and the diffusion update with in-kernel Philox.
// All loads/stores are 16 bytes per lane where the layout allows (fp32 rows of 512 = 2 x float4 per lane).
#include "common.h"
#include "tcdiff_hip.h"

// ---- fragment-ordered K / V cache images for the in-chain cross-attention (csrc/chain.hip cross_attention) -----------
// Element index of K[key][d] / V[key][d] inside the image of one (slot, head): 4 KB per 32-key tile, each 1-KB piece the 64
// lanes' 16-byte A-operand fragments of one v_mfma_f32_16x16x32_bf16 (lane = 16 g + c, 8 elements jj), in the k order in which
// the chain kernel's ACCUMULATORS hold the other operand (include/tcdiff_hip.h, tcdiff_pack_kv_frags):
//   K: piece (key % 32) / 16 * 2 + d / 32, c = key % 16, slot 8 g + jj <-> d % 32 = 16 (jj / 4) + 4 g + jj % 4
//   V: piece d / 16,                      c = d % 16,   slot 8 g + jj <-> key % 32 = 16 (jj / 4) + 4 g + jj % 4
DEVINL long kf_index(int key, int d) {
    const int kt = key >> 5, k32 = key & 31, d32 = d & 31;
    const int g = (d32 & 15) >> 2, jj = 4 * (d32 >> 4) + (d32 & 3);
    return ((long)((kt * 2 + (k32 >> 4)) * 2 + (d >> 5)) * 64 + g * 16 + (k32 & 15)) * 8 + jj;
}
DEVINL long vf_index(int key, int d) {
    const int kt = key >> 5, k32 = key & 31;
    const int g = (k32 & 15) >> 2, jj = 4 * (k32 >> 4) + (k32 & 3);
    return ((long)(kt * 4 + (d >> 4)) * 64 + g * 16 + (d & 15)) * 8 + jj;
}

// =================================================================================================
// LayerNorm (+ rotary): one wave per 512-wide row; lane l owns columns [4l, 4l+4) and [256+4l, 256+4l+4),
// so every rotary pair (2j, 2j+1) is inside one lane's float4 -- no cross-lane traffic for the rotation.
// (model/model.py:326,332,338,344 LayerNorm eps 1e-5; rotary model/rotary_embedding_torch.py:46-59)
// =================================================================================================
template <class P>
__global__ __launch_bounds__(256) void ln_rot_kernel(const float* __restrict__ x, int rows, const float* __restrict__ g,
                                                     const float* __restrict__ b, float eps, void* __restrict__ hout,
                                                     void* __restrict__ rout, float* __restrict__ y32,
                                                     const float* __restrict__ rope, int pos_mod, int pos_base) {
    typedef typename P::elem_t T;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + (long)row * 512;
    f32x4_t v[2];
    v[0] = *reinterpret_cast<const f32x4_t*>(xr + 4 * lane);
    v[1] = *reinterpret_cast<const f32x4_t*>(xr + 256 + 4 * lane);
    float s = (v[0].x + v[0].y) + (v[0].z + v[0].w) + (v[1].x + v[1].y) + (v[1].z + v[1].w);
    const float mean = wave_sum(s) * (1.0f / 512.0f);
    float ss = 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float d = v[i][j] - mean;
            ss += d * d;
        }
    const float rstd = rsqrtf(wave_sum(ss) * (1.0f / 512.0f) + eps);
    const int pos = pos_base + (pos_mod > 0 ? row % pos_mod : row);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c0 = i * 256 + 4 * lane;
        f32x4_t gg = *reinterpret_cast<const f32x4_t*>(g + c0);
        f32x4_t bb = *reinterpret_cast<const f32x4_t*>(b + c0);
        f32x4_t u;
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = (v[i][j] - mean) * rstd * gg[j] + bb[j];
        if (y32) *reinterpret_cast<f32x4_t*>(y32 + (long)row * 512 + c0) = u;
        if (hout) {
            T* hp = reinterpret_cast<T*>(hout) + (long)row * 512 + c0;
            if (P::IS_BF16) {
                uint2 pk;
                pk.x = pack_bf2(u[0], u[1]);
                pk.y = pack_bf2(u[2], u[3]);
                *reinterpret_cast<uint2*>(hp) = pk;
            } else {
                *reinterpret_cast<u32x4*>(hp) = P::chunk_from4(u);
            }
        }
        if (rout) {
            f32x4_t cs = *reinterpret_cast<const f32x4_t*>(rope + (long)pos * 512 + c0);  // cos0 sin0 cos1 sin1
            f32x4_t y;
            y[0] = u[0] * cs[0] - u[1] * cs[1];
            y[1] = u[1] * cs[0] + u[0] * cs[1];
            y[2] = u[2] * cs[2] - u[3] * cs[3];
            y[3] = u[3] * cs[2] + u[2] * cs[3];
            T* rp = reinterpret_cast<T*>(rout) + (long)row * 512 + c0;
            if (P::IS_BF16) {
                uint2 pk;
                pk.x = pack_bf2(y[0], y[1]);
                pk.y = pack_bf2(y[2], y[3]);
                *reinterpret_cast<uint2*>(rp) = pk;
            } else {
                *reinterpret_cast<u32x4*>(rp) = P::chunk_from4(y);
            }
        }
    }
}

extern "C" int tcdiff_ln_rot(int dtype, const float* x, int rows, const float* g, const float* b, float eps, void* h,
                             void* rot, float* y32, const float* rope, int pos_mod, int pos_base,
                             hipStream_t stream) {
    if (!x || !g || !b || rows <= 0 || (rot && !rope)) return TC_ERR_ARG;
    if (dtype != TC_DTYPE_BF16 && dtype != TC_DTYPE_F32 && dtype != TC_DTYPE_BF16X3) return TC_ERR_ARG;
    dim3 grid((rows + 3) / 4);
    if (dtype == TC_DTYPE_BF16)
        hipLaunchKernelGGL(ln_rot_kernel<MmaBF16>, grid, dim3(256), 0, stream, x, rows, g, b, eps, h, rot, y32, rope,
                           pos_mod, pos_base);
    else if (dtype == TC_DTYPE_BF16X3)
        hipLaunchKernelGGL(ln_rot_kernel<MmaBF16x3>, grid, dim3(256), 0, stream, x, rows, g, b, eps, h, rot, y32, rope,
                           pos_mod, pos_base);
    else
        hipLaunchKernelGGL(ln_rot_kernel<MmaF32>, grid, dim3(256), 0, stream, x, rows, g, b, eps, h, rot, y32, rope,
                           pos_mod, pos_base);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// rope[p][2j] = cos(p * freqs[j]), rope[p][2j+1] = sin(p * freqs[j]): angle = fp32 product as in the reference
// (model/rotary_embedding_torch.py:123), accurate sincosf (no fast-math).
__global__ void rope_table_kernel(const float* __restrict__ freqs, float* __restrict__ rope, int n_pos) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pos * 256) return;
    int p = i >> 8, j = i & 255;
    float ang = (float)p * freqs[j];
    float s, c;
    sincosf(ang, &s, &c);
    rope[(long)p * 512 + 2 * j] = c;
    rope[(long)p * 512 + 2 * j + 1] = s;
}

extern "C" int tcdiff_rope_table(const float* freqs, float* rope, int n_pos, hipStream_t stream) {
    if (!freqs || !rope || n_pos <= 0) return TC_ERR_ARG;
    int n = n_pos * 256;
    hipLaunchKernelGGL(rope_table_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, freqs, rope, n_pos);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// =================================================================================================
// conditioning-path helpers
// =================================================================================================
template <class P>
__global__ void convert_pad_kernel(const float* __restrict__ src, typename P::elem_t* __restrict__ dst, int rows,
                                   int cols, int ld_dst, int rows_per_batch, long batch_stride, long row_stride) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows * ld_dst) return;
    int r = (int)(i / ld_dst), c = (int)(i % ld_dst);
    float v = 0.0f;
    if (c < cols) v = src[(long)(r / rows_per_batch) * batch_stride + (long)(r % rows_per_batch) * row_stride + c];
    store_T<P>(dst, i, v);
}

extern "C" int tcdiff_convert_pad(int dtype, const float* src, void* dst, int rows, int cols, int ld_dst,
                                  int rows_per_batch, long batch_stride, long row_stride, hipStream_t stream) {
    if (!src || !dst || rows <= 0 || cols <= 0 || ld_dst < cols || rows_per_batch <= 0) return TC_ERR_ARG;
    long n = (long)rows * ld_dst;
    dim3 grid((unsigned)((n + 255) / 256));
    if (dtype == TC_DTYPE_BF16)
        hipLaunchKernelGGL(convert_pad_kernel<MmaBF16>, grid, dim3(256), 0, stream, src, (uint16_t*)dst, rows, cols,
                           ld_dst, rows_per_batch, batch_stride, row_stride);
    else if (dtype == TC_DTYPE_F32)
        hipLaunchKernelGGL(convert_pad_kernel<MmaF32>, grid, dim3(256), 0, stream, src, (float*)dst, rows, cols, ld_dst,
                           rows_per_batch, batch_stride, row_stride);
    else if (dtype == TC_DTYPE_BF16X3 && ld_dst % 4 == 0)
        hipLaunchKernelGGL(convert_pad_kernel<MmaBF16x3>, grid, dim3(256), 0, stream, src, (x3_word*)dst, rows, cols, ld_dst,
                           rows_per_batch, batch_stride, row_stride);
    else
        return TC_ERR_ARG;
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// SinusoidalPosEmb (model/utils.py:36-48): emb = [sin(t*f_k) | cos(t*f_k)], k < 256; f_k supplied by the host
// (a constant of the architecture, exp(-k ln(1e4)/255) evaluated as the reference does).
template <class P>
__global__ void sinusoidal_kernel(const int* __restrict__ times, int n, const float* __restrict__ freq,
                                  typename P::elem_t* __restrict__ emb) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 256) return;
    int row = i >> 8, k = i & 255;
    float a = (float)times[row] * freq[k];
    float s, c;
    sincosf(a, &s, &c);
    store_T<P>(emb, (long)row * 512 + k, s);
    store_T<P>(emb, (long)row * 512 + 256 + k, c);
}

extern "C" int tcdiff_sinusoidal(int dtype, const int* times, int n, const float* freq, void* emb,
                                 hipStream_t stream) {
    if (!times || !freq || !emb || n <= 0) return TC_ERR_ARG;
    dim3 grid((n * 256 + 255) / 256);
    if (dtype == TC_DTYPE_BF16)
        hipLaunchKernelGGL(sinusoidal_kernel<MmaBF16>, grid, dim3(256), 0, stream, times, n, freq, (uint16_t*)emb);
    else if (dtype == TC_DTYPE_F32)
        hipLaunchKernelGGL(sinusoidal_kernel<MmaF32>, grid, dim3(256), 0, stream, times, n, freq, (float*)emb);
    else if (dtype == TC_DTYPE_BF16X3)
        hipLaunchKernelGGL(sinusoidal_kernel<MmaBF16x3>, grid, dim3(256), 0, stream, times, n, freq, (x3_word*)emb);
    else
        return TC_ERR_ARG;
    TC_CHECK_LAUNCH();
    return TC_OK;
}

__global__ void mean_pool_kernel(const float* __restrict__ x, float* __restrict__ out, int B, int S, int C) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * C) return;
    int b = i / C, c = i % C;
    const float* p = x + (long)b * S * C + c;
    float s = 0.0f;
    for (int t = 0; t < S; ++t) s += p[(long)t * C];
    out[i] = s / (float)S;
}

extern "C" int tcdiff_mean_pool(const float* x, float* out, int B, int S, int C, hipStream_t stream) {
    if (!x || !out || B <= 0 || S <= 0 || C <= 0) return TC_ERR_ARG;
    hipLaunchKernelGGL(mean_pool_kernel, dim3((B * C + 255) / 256), dim3(256), 0, stream, x, out, B, S, C);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

template <class P>
__global__ void add_act_kernel(const float* __restrict__ a, const int* __restrict__ ia, const float* __restrict__ b,
                               int n, int act, typename P::elem_t* __restrict__ out, float* __restrict__ out32) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * 512) return;
    int row = i >> 9, c = i & 511;
    int ra = ia ? ia[row] : row;
    float v = a[(long)ra * 512 + c] + (b ? b[i] : 0.0f);
    v = apply_act(v, act);
    if (out) store_T<P>(out, i, v);
    if (out32) out32[i] = v;
}

extern "C" int tcdiff_add_act(int dtype, const float* a, const int* ia, const float* b, int n, int act, void* out,
                              float* out32, hipStream_t stream) {
    if (!a || n <= 0 || (!out && !out32)) return TC_ERR_ARG;
    dim3 grid((n * 512 + 255) / 256);
    if (dtype == TC_DTYPE_BF16)
        hipLaunchKernelGGL(add_act_kernel<MmaBF16>, grid, dim3(256), 0, stream, a, ia, b, n, act, (uint16_t*)out, out32);
    else if (dtype == TC_DTYPE_F32)
        hipLaunchKernelGGL(add_act_kernel<MmaF32>, grid, dim3(256), 0, stream, a, ia, b, n, act, (float*)out, out32);
    else if (dtype == TC_DTYPE_BF16X3)
        hipLaunchKernelGGL(add_act_kernel<MmaBF16x3>, grid, dim3(256), 0, stream, a, ia, b, n, act, (x3_word*)out, out32);
    else
        return TC_ERR_ARG;
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// per-step: the two time-token rows of every layer's cross-attention K / V cache
template <class P>
__global__ void scatter_time_kv_kernel(const typename P::elem_t* __restrict__ tab, int n_t,
                                       const int* __restrict__ tidx, typename P::elem_t* __restrict__ Kc,
                                       typename P::elem_t* __restrict__ Vc, int NL, int n_kv, int H, int Lp,
                                       int tok0) {
    // one thread per (layer, kv, row r in {0,1}, column c in [0,1024))
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)NL * n_kv * 2 * 1024;
    if (i >= total) return;
    int c = (int)(i & 1023);
    int rr = (int)((i >> 10) & 1);
    long rest = i >> 11;
    int kv = (int)(rest % n_kv), l = (int)(rest / n_kv);
    int t = tidx[kv];
    typename P::elem_t v = tab[(((long)l * n_t + t) * 2 + rr) * 1024 + c];
    int tok = tok0 + rr;
    const int cc = c & 511;
    const int head = cc >> 6, d = cc & 63;
    typename P::elem_t* dst = c < 512 ? Kc : Vc;
    dst[((((long)l * n_kv + kv) * H + head) * Lp + tok) * 64 + d] = v;
}

extern "C" int tcdiff_scatter_time_kv(int dtype, const void* tab, int n_t, const int* tidx, void* Kc, void* Vc,
                                      int NL, int n_kv, int H, int Lp, int tok0, hipStream_t stream) {
    if (!tab || !tidx || !Kc || !Vc || NL <= 0 || n_kv <= 0 || H * 64 != 512 || tok0 + 1 >= Lp) return TC_ERR_ARG;
    long total = (long)NL * n_kv * 2 * 1024;
    dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == TC_DTYPE_BF16)
        hipLaunchKernelGGL(scatter_time_kv_kernel<MmaBF16>, grid, dim3(256), 0, stream, (const uint16_t*)tab, n_t, tidx,
                           (uint16_t*)Kc, (uint16_t*)Vc, NL, n_kv, H, Lp, tok0);
    else if (dtype == TC_DTYPE_F32)
        hipLaunchKernelGGL(scatter_time_kv_kernel<MmaF32>, grid, dim3(256), 0, stream, (const float*)tab, n_t, tidx,
                           (float*)Kc, (float*)Vc, NL, n_kv, H, Lp, tok0);
    else if (dtype == TC_DTYPE_BF16X3)   // 4-byte slots copied one by one: a split chunk's four slots move together (d = c mod 64)
        hipLaunchKernelGGL(scatter_time_kv_kernel<MmaF32>, grid, dim3(256), 0, stream, (const float*)tab, n_t, tidx,
                           (float*)Kc, (float*)Vc, NL, n_kv, H, Lp, tok0);
    else
        return TC_ERR_ARG;
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// =================================================================================================
// sampler: device-resident step state so that ONE captured graph serves every step
// =================================================================================================
__global__ void step_begin_kernel(const int* __restrict__ counter, const int* __restrict__ tseq, int* __restrict__ tidx,
                                  int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tidx[i] = tseq[counter[0]];
}
__global__ void step_end_kernel(int* counter) {
    if (threadIdx.x == 0 && blockIdx.x == 0) counter[0] += 1;
}

extern "C" int tcdiff_step_begin(const int* counter, const int* tseq, int* tidx, int n, hipStream_t stream) {
    if (!counter || !tseq || !tidx || n <= 0) return TC_ERR_ARG;
    hipLaunchKernelGGL(step_begin_kernel, dim3((n + 63) / 64), dim3(64), 0, stream, counter, tseq, tidx, n);
    TC_CHECK_LAUNCH();
    return TC_OK;
}
extern "C" int tcdiff_step_end(int* counter, hipStream_t stream) {
    if (!counter) return TC_ERR_ARG;
    hipLaunchKernelGGL(step_end_kernel, dim3(1), dim3(64), 0, stream, counter);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// One launch for everything a sampler step does before the network: timestep lookup, FiLM generator input
// (mish(time_base[t] + hidden[row])), the two time-token K / V rows of every layer (row-major cache and / or the
// fragment-ordered images chain.hip reads), the model-dtype copy of x_t, and the step counter bump.
// counter = {current step (written here), seed0, seed1, next step}: this launch only READS counter[3] and only WRITES
// counter[0]; sampler_update (mode | TC_SAMPLER_ADVANCE) only reads counter[0] and writes counter[3] = step + 1.  The
// kernel boundary between the two orders everything: no atomics, no device-scope fence (on gfx950 an agent-scope
// release writes the XCD's L2 back; a ticket counter built on one cost 45 us per launch here).
template <class P>
__global__ __launch_bounds__(256) void step_prologue_kernel(tcdiff_step_prologue_args a) {
    typedef typename P::elem_t E;
    int* counter = a.counter;
    const int step = counter[3];
    const int t = a.tseq[step];
    // index ranges of the launch's parts (tcdiff_step_prologue_args.parts): [FiLM input / tidx | time-token rows | x copy | FiLM gather]
    const bool do_x = a.parts == 0 || (a.parts & TC_PROLOGUE_X), do_c = a.parts == 0 || (a.parts & TC_PROLOGUE_COND);
    const long n_film = (long)a.n_seq * 512;
    const long n_kv = do_c ? (long)a.NL * a.n_kv * 2 * 1024 : 0;
    // the x copy: bf16 with whole 16-byte chunks per row moves eight elements per thread (one 16-byte store instead of eight 2-byte
    // ones: this part is two thirds of the launch's items at the benchmarked shape)
    const bool x8 = std::is_same<P, MmaBF16>::value && a.ld_xin % 8 == 0;
    const long n_x = (a.x && do_x) ? (x8 ? (long)a.rows * (a.ld_xin / 8) : (long)a.rows * a.ld_xin) : 0;
    const long q_tab = (a.film_tab && do_c) ? (long)a.n_seq * (a.nfilm / 4) : 0;       // float4 pieces of the gathered FiLM rows
    const long total = n_film + n_kv + n_x + q_tab;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        if (i >= n_film + n_kv + n_x) {
            const long k = i - (n_film + n_kv + n_x);
            const int q4 = a.nfilm / 4;
            const int seq = (int)(k / q4), c4 = (int)(k % q4);
            const int j = seq < a.n_unc ? 0 : seq - a.n_unc + 1;
            reinterpret_cast<f32x4_t*>(a.film_out)[(long)seq * q4 + c4] =
                reinterpret_cast<const f32x4_t*>(a.film_tab)[((long)t * a.film_rows + j) * q4 + c4];
        } else if (i < n_film) {
            const int c = (int)(i & 511);
            if (c == 0 && do_x) a.tidx[i >> 9] = t;
            if (!a.film_tab && do_c) store_T<P>((E*)a.film_in, i, mish_f(a.t_base[(long)t * 512 + c] + a.hidden[i]));
        } else if (i < n_film + n_kv) {
            const long k = i - n_film;
            const int c = (int)(k & 1023), rr = (int)((k >> 10) & 1);
            const long rest = k >> 11;
            const int kv = (int)(rest % a.n_kv), l = (int)(rest / a.n_kv);
            const E v = ((const E*)a.tab)[(((long)l * a.n_t + t) * 2 + rr) * 1024 + c];
            const int tok = a.tok0 + rr, cc = c & 511, head = cc >> 6, d = cc & 63;
            const long sh = ((long)l * a.n_kv + kv) * a.H + head;
            if (a.Kc) ((E*)(c < 512 ? a.Kc : a.Vc))[(sh * a.Lp + tok) * 64 + d] = v;
            if (a.Kf) {  // same element maps as pack_kv_frags_kernel
                const long base = sh * (long)a.nkt * 2048;
                if (c < 512) ((E*)a.Kf)[base + kf_index(tok, d)] = v;
                else ((E*)a.Vf)[base + vf_index(tok, d)] = v;
            }
        } else {
            const long k = i - n_film - n_kv;
            if (x8) {
                const int q8 = a.ld_xin / 8;
                const int r = (int)(k / q8), c0 = (int)(k % q8) * 8;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = c0 + j < a.nfeat ? a.x[(long)r * a.nfeat + c0 + j] : 0.0f;
                const u32x4 pk = {pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
                *reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(a.xin) + (long)r * a.ld_xin + c0) = pk;
            } else {
                const int r = (int)(k / a.ld_xin), c = (int)(k % a.ld_xin);
                store_T<P>((E*)a.xin, k, c < a.nfeat ? a.x[(long)r * a.nfeat + c] : 0.0f);
            }
        }
    }
    if (do_x && blockIdx.x == 0 && threadIdx.x == 0) counter[0] = step;
}

extern "C" int tcdiff_step_prologue(int dtype, const tcdiff_step_prologue_args* a, hipStream_t stream) {
    if (!a || !a->counter || !a->tseq || !a->tidx || !a->t_base || !a->hidden || !a->film_in || !a->tab ||
        a->n_seq <= 0 || a->NL <= 0 || a->n_kv <= 0 || a->H * 64 != 512 || (!a->Kc && !a->Kf) || (!a->Kc != !a->Vc) ||
        (!a->Kf != !a->Vf) || a->tok0 + 1 >= a->Lp || (a->Kf && a->tok0 + 2 > 32 * a->nkt) ||
        (a->x && (!a->xin || a->rows <= 0 || a->ld_xin < a->nfeat)))
        return TC_ERR_ARG;
    if (a->Kf && dtype != TC_DTYPE_BF16) return TC_ERR_ARG;
    if (a->film_tab && (!a->film_out || a->film_rows <= 0 || a->nfilm <= 0 || a->nfilm % 4 || a->n_unc < 0 ||
                        a->n_seq - a->n_unc + 1 > a->film_rows ||
                        ((reinterpret_cast<uintptr_t>(a->film_tab) | reinterpret_cast<uintptr_t>(a->film_out)) & 15)))
        return TC_ERR_ARG;
    if (a->parts < 0 || a->parts > (TC_PROLOGUE_X | TC_PROLOGUE_COND)) return TC_ERR_ARG;
    const bool do_x = a->parts == 0 || (a->parts & TC_PROLOGUE_X), do_c = a->parts == 0 || (a->parts & TC_PROLOGUE_COND);
    const bool x8 = dtype == TC_DTYPE_BF16 && a->ld_xin % 8 == 0;
    if (x8 && a->x && (reinterpret_cast<uintptr_t>(a->xin) & 15)) return TC_ERR_ALIGN;
    const long total = (long)a->n_seq * 512 + (do_c ? (long)a->NL * a->n_kv * 2048 : 0) +
                       ((a->x && do_x) ? (x8 ? (long)a->rows * (a->ld_xin / 8) : (long)a->rows * a->ld_xin) : 0) +
                       ((a->film_tab && do_c) ? (long)a->n_seq * (a->nfilm / 4) : 0);
    const unsigned grid = (unsigned)std::min<long>((total + 255) / 256, 2048);
    if (dtype == TC_DTYPE_BF16)
        hipLaunchKernelGGL(step_prologue_kernel<MmaBF16>, dim3(grid), dim3(256), 0, stream, *a);
    else if (dtype == TC_DTYPE_F32)
        hipLaunchKernelGGL(step_prologue_kernel<MmaF32>, dim3(grid), dim3(256), 0, stream, *a);
    else if (dtype == TC_DTYPE_BF16X3 && a->ld_xin % 4 == 0)
        hipLaunchKernelGGL(step_prologue_kernel<MmaBF16x3>, dim3(grid), dim3(256), 0, stream, *a);
    else
        return TC_ERR_ARG;
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// Philox4x32-10 (Salmon et al. 2011), counter = (element quad, timestep, clip, 0), key = seed
DEVINL void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}
DEVINL void philox4(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
DEVINL float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)

__global__ void sampler_update_kernel(int mode, const float* __restrict__ out_unc, const float* __restrict__ out_cond,
                                      int ldo, float* __restrict__ x, const float* __restrict__ eps,
                                      const float* __restrict__ traj, float* __restrict__ x0_out, int n_rows,
                                      int nfeat, int L, int* counter,
                                      const float* __restrict__ params, const int* __restrict__ tseq, uint64_t seed,
                                      int clip0, int advance) {
    // one thread per 4 consecutive elements of a row (so one Philox call feeds 4 normals)
    const int quads = (nfeat + 3) / 4;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n_rows * quads) return;
    const int row = (int)(i / quads), qd = (int)(i % quads);
    const int step = counter[0];
    if (advance && i == 0) counter[3] = step + 1;                      // see step_prologue_kernel
    const float* pr = params + (long)step * 8;
    const float w = pr[0];
    const int flags = (int)pr[7];
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (!eps) {
        const int clip = clip0 + row / L, tok = row % L;
        uint32_t c[4] = {(uint32_t)(tok * quads + qd), (uint32_t)tseq[step], (uint32_t)clip, 0u};
        // counter[1], counter[2]: device-side seed words (XORed into the by-value seed) so that a captured
        // graph can be replayed with a new seed
        philox4(c, (uint32_t)seed ^ (uint32_t)counter[1], (uint32_t)(seed >> 32) ^ (uint32_t)counter[2]);
        float u0 = u01(c[0]), u1 = u01(c[1]), u2 = u01(c[2]), u3 = u01(c[3]);
        float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
        float s0, c0, s1, c1;
        sincosf(6.283185307179586f * u1, &s0, &c0);
        sincosf(6.283185307179586f * u3, &s1, &c1);
        z[0] = r0 * c0; z[1] = r0 * s0; z[2] = r1 * c1; z[3] = r1 * s1;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = qd * 4 + j;
        if (c >= nfeat) break;
        const long xi = (long)row * nfeat + c;
        const float oc = out_cond[(long)row * ldo + c];
        float g;
        if (out_unc) {
            const float ou = out_unc[(long)row * ldo + c];
            g = ou + (oc - ou) * w;  // model/model.py:546
        } else {
            g = oc;
        }
        const float xt = x[xi];
        // params[7] bit 2: the network predicts the noise, x_0 = sqrt(1/ac) x_t - sqrt(1/ac - 1) eps_hat (predict_epsilon,
        // model/diffusion.py:176-187; DDPM steps only: model_predictions :195-204 takes the output as x_0 either way);
        // bit 3: no clamp (clip_denoised=False: :230-233 for DDPM, and the three DDIM samplers pass
        // clip_x_start=self.clip_denoised, :316,409,476)
        if (mode == TC_SAMPLER_DDPM && (flags & 4)) g = pr[4] * xt - pr[5] * g;
        const float x0 = (flags & 8) ? g : fminf(fmaxf(g, -1.0f), 1.0f);  // :230-231 / :199-201
        const float e = eps ? eps[xi] : z[j];
        float xn;
        if (mode == TC_SAMPLER_DDPM) {
            // model/diffusion.py:207-210,251: mean = coef1*x0 + coef2*x_t ; x = mean + sigma*eps (sigma = 0 at t = 0)
            xn = (pr[1] * x0 + pr[2] * xt) + pr[3] * e;
        } else {
            // model/diffusion.py:189-193,421-425
            const float pn = (pr[1] * xt - x0) / pr[2];
            xn = pr[6] != 0.0f ? x0 : (x0 * pr[3] + pr[4] * pn) + pr[5] * e;
        }
        if (traj && (c == 4 || c == 5)) xn = traj[(long)row * 3 + (c - 4)];  // model/diffusion.py:427-431
        x[xi] = xn;
        if (x0_out) x0_out[xi] = x0;
    }
}

extern "C" int tcdiff_sampler_update(int mode, const float* out_unc, const float* out_cond, int ldo, float* x,
                                     const float* eps, const float* traj, float* x0_out, int n_rows, int nfeat, int L,
                                     const int* counter, const float* params, const int* tseq, uint64_t seed, int clip0,
                                     hipStream_t stream) {
    if (!out_cond || !x || !counter || !params || !tseq || n_rows <= 0 || nfeat <= 0 || L <= 0 || ldo < nfeat)
        return TC_ERR_ARG;
    const int advance = (mode & TC_SAMPLER_ADVANCE) != 0;
    mode &= ~TC_SAMPLER_ADVANCE;
    if (mode != TC_SAMPLER_DDPM && mode != TC_SAMPLER_DDIM) return TC_ERR_ARG;
    long n = (long)n_rows * ((nfeat + 3) / 4);
    hipLaunchKernelGGL(sampler_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, mode, out_unc,
                       out_cond, ldo, x, eps, traj, x0_out, n_rows, nfeat, L, const_cast<int*>(counter), params, tseq, seed,
                       clip0, advance);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// x[1:, :half] = x[:-1, half:] (model/diffusion.py:502-506).  Source and destination ranges of one clip never
// overlap (first half vs second half), and clip i's second half is only read, so one pass is race-free.
__global__ void window_couple_kernel(float* __restrict__ x, int b, long half_elems) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)(b - 1) * half_elems) return;
    long clip = 1 + i / half_elems, e = i % half_elems;
    x[clip * 2 * half_elems + e] = x[(clip - 1) * 2 * half_elems + half_elems + e];
}

extern "C" int tcdiff_window_couple(float* x, int b, int seq_len, int row_elems, hipStream_t stream) {
    if (!x || b <= 0 || seq_len <= 0 || (seq_len & 1) || row_elems <= 0) return TC_ERR_ARG;
    if (b == 1) return TC_OK;
    long half = (long)(seq_len / 2) * row_elems;
    long n = (long)(b - 1) * half;
    hipLaunchKernelGGL(window_couple_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, b, half);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

__global__ void pack_kv_frags_kernel(const uint16_t* __restrict__ Kc, const uint16_t* __restrict__ Vc,
                                     uint16_t* __restrict__ Kf, uint16_t* __restrict__ Vf, long n_sh, int Lp, int nkt,
                                     int key_lo, int nkeys) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // (slot*head, key, d)
    if (i >= n_sh * nkeys * 64) return;
    const int d = (int)(i & 63);
    const long r2 = i >> 6;
    const int key = key_lo + (int)(r2 % nkeys);
    const long sh = r2 / nkeys;
    const long src = (sh * Lp + key) * 64 + d;
    const long base = sh * (long)nkt * 2048;                              // elements per (slot, head) image
    Kf[base + kf_index(key, d)] = Kc[src];
    Vf[base + vf_index(key, d)] = Vc[src];
}

extern "C" int tcdiff_pack_kv_frags(const void* Kc, const void* Vc, void* Kf, void* Vf, int n_slots, int H, int Lp,
                                    int nkt, int key_lo, int key_hi, hipStream_t stream) {
    if (!Kc || !Vc || !Kf || !Vf || n_slots <= 0 || H <= 0 || Lp <= 0 || nkt <= 0 || key_lo < 0 || key_hi <= key_lo ||
        key_hi > 32 * nkt || key_hi > Lp)
        return TC_ERR_ARG;
    const long n = (long)n_slots * H * (key_hi - key_lo) * 64;
    hipLaunchKernelGGL(pack_kv_frags_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream,
                       (const uint16_t*)Kc, (const uint16_t*)Vc, (uint16_t*)Kf, (uint16_t*)Vf, (long)n_slots * H, Lp, nkt,
                       key_lo, key_hi - key_lo);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// ---- in-painting constraints and window coupling INSIDE the captured step (no per-step host callback) ---------------
// kind 1: x = mask ? value : x while the step is not the last one (ddim_sample_Footwork, model/diffusion.py:341-356)
// kind 2: x = q_sample(value, t - 1) * mask + (1 - mask) * x for t > 0 (inpaint_loop, model/diffusion.py:545-551);
//         q_sample = params[4] * value + params[5] * noise, noise injected (q_eps) or Philox (stream word 1)
__global__ void sampler_constrain_kernel(int kind, float* __restrict__ x, const float* __restrict__ mask, int mask_rows,
                                         const float* __restrict__ value, const float* __restrict__ q_eps, int n_rows,
                                         int nfeat, int L, const int* __restrict__ counter,
                                         const float* __restrict__ params, const int* __restrict__ tseq, uint64_t seed,
                                         int clip0) {
    const int quads = (nfeat + 3) / 4;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n_rows * quads) return;
    const int row = (int)(i / quads), qd = (int)(i % quads);
    const int step = counter[0];
    const float* pr = params + (long)step * 8;
    if ((((int)pr[7]) & 2) == 0) return;           // column 7 bit 1: constraint enabled for this step (host-side schedule)
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (kind == 2 && !q_eps) {
        const int clip = clip0 + row / L, tok = row % L;
        uint32_t c[4] = {(uint32_t)(tok * quads + qd), (uint32_t)tseq[step], (uint32_t)clip, 1u};
        philox4(c, (uint32_t)seed ^ (uint32_t)counter[1], (uint32_t)(seed >> 32) ^ (uint32_t)counter[2]);
        const float u0 = u01(c[0]), u1 = u01(c[1]), u2 = u01(c[2]), u3 = u01(c[3]);
        const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
        float s0, c0, s1, c1;
        sincosf(6.283185307179586f * u1, &s0, &c0);
        sincosf(6.283185307179586f * u3, &s1, &c1);
        z[0] = r0 * c0; z[1] = r0 * s0; z[2] = r1 * c1; z[3] = r1 * s1;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = qd * 4 + j;
        if (c >= nfeat) break;
        const long xi = (long)row * nfeat + c;
        const float m = mask[(long)(row % mask_rows) * nfeat + c];
        const float v = value[xi];
        if (kind == 1) {
            if (m != 0.0f) x[xi] = v;
        } else {
            const float e = q_eps ? q_eps[xi] : z[j];
            const float qs = pr[4] * v + pr[5] * e;
            x[xi] = qs * m + (1.0f - m) * x[xi];
        }
    }
}

extern "C" int tcdiff_sampler_constrain(int kind, float* x, const float* mask, int mask_rows, const float* value,
                                        const float* q_eps, int n_rows, int nfeat, int L, const int* counter,
                                        const float* params, const int* tseq, uint64_t seed, int clip0,
                                        hipStream_t stream) {
    if ((kind != 1 && kind != 2) || !x || !mask || !value || !counter || !params || !tseq || n_rows <= 0 || nfeat <= 0 ||
        L <= 0 || mask_rows <= 0)
        return TC_ERR_ARG;
    long n = (long)n_rows * ((nfeat + 3) / 4);
    hipLaunchKernelGGL(sampler_constrain_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, kind, x, mask,
                       mask_rows, value, q_eps, n_rows, nfeat, L, counter, params, tseq, seed, clip0);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// window coupling gated by the step's flag column 7 (the reference skips it after the last step)
__global__ void window_couple_step_kernel(float* __restrict__ x, int b, long half_elems, const int* __restrict__ counter,
                                          const float* __restrict__ params) {
    if ((((int)params[(long)counter[0] * 8 + 7]) & 1) == 0) return;   // column 7 bit 0: couple after this step
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)(b - 1) * half_elems) return;
    long clip = 1 + i / half_elems, e = i % half_elems;
    x[clip * 2 * half_elems + e] = x[(clip - 1) * 2 * half_elems + half_elems + e];
}

extern "C" int tcdiff_window_couple_step(float* x, int b, int seq_len, int row_elems, const int* counter,
                                         const float* params, hipStream_t stream) {
    if (!x || !counter || !params || b <= 0 || seq_len <= 0 || (seq_len & 1) || row_elems <= 0) return TC_ERR_ARG;
    if (b == 1) return TC_OK;
    long half = (long)(seq_len / 2) * row_elems;
    long n = (long)(b - 1) * half;
    hipLaunchKernelGGL(window_couple_step_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, b, half,
                       counter, params);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

__global__ void cfg_combine_kernel(const float* __restrict__ ou, const float* __restrict__ oc, int ldo, float w,
                                   float* __restrict__ y, int n_rows, int nfeat) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n_rows * nfeat) return;
    int row = (int)(i / nfeat), c = (int)(i % nfeat);
    float u = ou[(long)row * ldo + c], v = oc[(long)row * ldo + c];
    y[i] = u + (v - u) * w;
}

extern "C" int tcdiff_cfg_combine(const float* out_unc, const float* out_cond, int ldo, float w, float* y, int n_rows,
                                  int nfeat, hipStream_t stream) {
    if (!out_unc || !out_cond || !y || n_rows <= 0 || nfeat <= 0 || ldo < nfeat) return TC_ERR_ARG;
    long n = (long)n_rows * nfeat;
    hipLaunchKernelGGL(cfg_combine_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, out_unc, out_cond,
                       ldo, w, y, n_rows, nfeat);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

// ---- EMA (model/diffusion.py:61-76): one workgroup per chunk of <= 65536 elements ----------------------------------
__global__ __launch_bounds__(256) void ema_update_kernel(const tcdiff_ema_chunk* __restrict__ chunks, float beta,
                                                         float omb) {
#pragma clang fp contract(off)   // the reference rounds both products before the sum: no fma (HIP's __fmul_rn is a plain *)
    const tcdiff_ema_chunk c = chunks[blockIdx.x];
    float* ma = c.ma;
    const float* cur = c.cur;
    const long n = c.n;
    const bool vec = ((reinterpret_cast<uintptr_t>(ma) | reinterpret_cast<uintptr_t>(cur)) & 15) == 0;
    long i0 = 0;
    if (vec) {
        const long n4 = n >> 2;
        for (long i = threadIdx.x; i < n4; i += 256) {
            f32x4_t a = reinterpret_cast<const f32x4_t*>(ma)[i];
            const f32x4_t b = reinterpret_cast<const f32x4_t*>(cur)[i];
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] = a[t] * beta + omb * b[t];
            reinterpret_cast<f32x4_t*>(ma)[i] = a;
        }
        i0 = n4 << 2;
    }
    for (long i = i0 + threadIdx.x; i < n; i += 256) ma[i] = ma[i] * beta + omb * cur[i];
}

extern "C" int tcdiff_ema_update(const tcdiff_ema_chunk* chunks, int n_chunks, float beta, float one_minus_beta,
                                 hipStream_t stream) {
    if (!chunks || n_chunks <= 0) return TC_ERR_ARG;
    hipLaunchKernelGGL(ema_update_kernel, dim3(n_chunks), dim3(256), 0, stream, chunks, beta, one_minus_beta);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

extern "C" const char* tcdiff_version(void) { return "tcdiff-gfx950 0.1 (bf16 32x32x16 / f32 32x32x2 MFMA)"; }
