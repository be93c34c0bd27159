// Train-mode attention of the TCDiff denoiser, forward and backward, gfx950 (d_k = 64 per head).
//
// Forward = attention.hip's streaming kernel plus (a) dropout on the softmax weights -- SBI_MSA.dropout on
// F.softmax(...) at model/model.py:98 and nn.MultiheadAttention(dropout=0.1)'s weights at :190-192,228-236 -- and
// (b) the row statistic lse = log2 sum_k 2^(s_k log2 e) the backward needs.
//
// Backward = what torch autograd derives for model/model.py:97-102 (softmax, dropout, two matmuls), flash-style: the
// L x L weights are never stored; P is recomputed from Q, K and lse, the dropout bits from the counter hash of
// train_common.h.  With S = Q' K^T (Q' already scaled by 1/sqrt(d_k)), P = softmax(S), Pd = mask P / (1 - p), O = Pd V:
//     dV = Pd^T dO            dPd = dO V^T          dP = mask dPd / (1 - p)
//     dS = P o (dP - delta),  delta_i = sum_j P_ij dP_ij = sum_d dO_id O_id
//     dQ' = dS K              dK = dS^T Q'
// Two launches with the forward's structure (operand roles swapped so that every product sums over the accumulator ROW
// index and its result tile is directly the next MFMA's B operand -- cdna_hip_programming.md section 3):
//   * dq kernel, query-major (a wave owns 32 queries, K / V tiles stream through LDS):
//       S^T = K Q'^T, dP^T = V dO^T (key in registers, query on the lane: lse and delta are per-lane scalars),
//       dQ'^T += K^T dS^T with K^T fragments read transposed from the staged K tile;
//   * dkv kernel, key-major (a wave owns 32 keys, Q' / dO tiles stream through LDS):
//       S = Q' K^T, dP = dO V^T (query in registers, key on the lane), dV^T += dO^T Pd, dK^T += Q'^T dS with dO^T / Q'^T
//       fragments read transposed from the staged tiles.
// S and dP are evaluated in both (7 MFMA products instead of 5) in exchange for no atomics: every output element is
// written once, by one wave, in a fixed summation order -- bitwise reproducible gradients.
#include "attn_res.h"        // attention_res_kernel<NG, TRAIN>: the K/V-resident forward
#include "tcdiff_hip.h"

#ifndef TC_DKV_NG
#define TC_DKV_NG 1      // 32-key groups per wave of the resident dK / dV kernel (2 would need ~280 VGPRs: spills)
#endif

// the B operand of the second product from an accumulator tile: k-step `st` of a 32-row tile (see attention.hip)
template <class P>
DEVINL u32x4 pack_frag(const f32x16_t& s, int st) {
    u32x4 pf;
    if (P::IS_BF16) {
        pf[0] = pack_bf2(s[8 * st + 0], s[8 * st + 1]);
        pf[1] = pack_bf2(s[8 * st + 2], s[8 * st + 3]);
        pf[2] = pack_bf2(s[8 * st + 4], s[8 * st + 5]);
        pf[3] = pack_bf2(s[8 * st + 6], s[8 * st + 7]);
    } else {
        const f32x4_t pv = {s[4 * st + 0], s[4 * st + 1], s[4 * st + 2], s[4 * st + 3]};
        pf = __builtin_bit_cast(u32x4, pv);
    }
    return pf;
}

// acc (+)= T[row0 + r][:] . regs[:]  for a staged [KB][64] tile T (rows on the accumulator ROW index)
template <class P>
DEVINL void mma_tile_rows(f32x16_t& acc, const char* tile, int row0, const u32x4* regs, int r, int h) {
    typedef AttnCfg<P> C;
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks) {
        const int sub = ks >> 2, ch = 2 * (ks & 3) + h;
        const u32x4 f = *reinterpret_cast<const u32x4*>(tile + sub * (C::KB * TC_ROWB) + tile_off(row0 + r, ch));
        P::mma(acc, f, regs[ks]);
    }
}

template <class P>
struct TileStager {   // global -> registers -> LDS staging of two [KB][64] tiles (the forward kernel's scheme)
    typedef AttnCfg<P> C;
    u32x4 a[2], b[2];
    DEVINL void load(const char* ga, const char* gb, int row0, int tid) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * 256;
            a[i] = *reinterpret_cast<const u32x4*>(ga + (long)row0 * 64 * C::ES + c * 16);
            b[i] = *reinterpret_cast<const u32x4*>(gb + (long)row0 * 64 * C::ES + c * 16);
        }
    }
    DEVINL void store(char* stage, int tid) const {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * 256;
            const int row = c / (8 * C::DSUB), chk = c % (8 * C::DSUB);
            const int off = (chk >> 3) * (C::KB * TC_ROWB) + tile_off(row, chk & 7);
            *reinterpret_cast<u32x4*>(stage + off) = a[i];
            *reinterpret_cast<u32x4*>(stage + C::TILE_BYTES + off) = b[i];
        }
    }
};

template <class P>
DEVINL void store_row4(void* base, long elem_off, float v0, float v1, float v2, float v3) {
    typedef typename P::elem_t T;
    T* p = reinterpret_cast<T*>(base) + elem_off;
    if (P::IS_BF16) {
        uint2 pk;
        pk.x = pack_bf2(v0, v1);
        pk.y = pack_bf2(v2, v3);
        *reinterpret_cast<uint2*>(p) = pk;
    } else {
        const f32x4_t pk = {v0, v1, v2, v3};
        *reinterpret_cast<f32x4_t*>(p) = pk;
    }
}

// =====================================================================================================================
// forward
// =====================================================================================================================
template <class P>
__global__ __launch_bounds__(256) void attention_train_kernel(const char* __restrict__ Q, const char* __restrict__ K,
                                                              const char* __restrict__ V, char* __restrict__ O,
                                                              char* __restrict__ O_lo, float* __restrict__ lse, int H, int Lq,
                                                              int Lk, int Lp_q,
                                                              int Lp_k, int ldo, const int* __restrict__ seed, int site,
                                                              uint32_t thr, float dscale) {
    typedef AttnCfg<P> C;
    constexpr int ES = C::ES, KB = C::KB;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nqb = Lp_q / 128;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int qblk = (wg % nqb) * 128;
    const int head = (wg / nqb) % H, seq = wg / (nqb * H);
    if (qblk >= Lq) return;
    const int q0 = qblk + wave * 32;
    const bool active = q0 < Lq;   // wave-uniform
    const int bh = seq * H + head;
    const DropCtx dc = drop_ctx(seed, site, thr, dscale);
    const uint32_t xrow = ((uint32_t)bh * (uint32_t)Lq + (uint32_t)(q0 + r)) * (uint32_t)Lk;   // flat index of (bh, q, 0)

    const char* Qg = Q + ((long)bh * Lp_q + q0 + r) * 64 * ES;
    const char* Kg = K + (long)bh * Lp_k * 64 * ES;
    const char* Vg = V + (long)bh * Lp_k * 64 * ES;
    u32x4 qf[C::NKS];
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks) qf[ks] = *reinterpret_cast<const u32x4*>(Qg + (2 * ks + h) * 16);

    TileStager<P> stg;
    f32x16_t o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int q = 0; q < 16; ++q) o[dt][q] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    const int nb = (Lk + KB - 1) / KB;
    stg.load(Kg, Vg, 0, tid);
    stg.store(smem, tid);
    __syncthreads();
    for (int b = 0; b < nb; ++b) {
        const int cur = b & 1, kv0 = b * KB;
        if (b + 1 < nb) stg.load(Kg, Vg, kv0 + KB, tid);
        const char* kt_base = smem + cur * C::STAGE;
        const char* vt_base = kt_base + C::TILE_BYTES;
        if (active) {
            f32x16_t s[C::NKT];
#pragma unroll
            for (int kt = 0; kt < C::NKT; ++kt) {
#pragma unroll
                for (int q = 0; q < 16; ++q) s[kt][q] = 0.0f;
                mma_tile_rows<P>(s[kt], kt_base, kt * 32, qf, r, h);
            }
            if (kv0 + KB > Lk) {
#pragma unroll
                for (int kt = 0; kt < C::NKT; ++kt)
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        if (kv0 + kt * 32 + acc_row(q, h) >= Lk) s[kt][q] = -INFINITY;
            }
            constexpr float LOG2E = 1.4426950408889634f;
            float mx = s[0][0];
#pragma unroll
            for (int kt = 0; kt < C::NKT; ++kt)
#pragma unroll
                for (int q = 0; q < 16; ++q) mx = fmaxf(mx, s[kt][q]);
            mx = fmaxf(mx, other_half(mx)) * LOG2E;
            const float m_new = fmaxf(m_run, mx);
            float rs = 0.0f;
#pragma unroll
            for (int kt = 0; kt < C::NKT; ++kt)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float p = __builtin_amdgcn_exp2f(fmaf(s[kt][q], LOG2E, -m_new));
                    rs += p;                                    // the softmax denominator is taken BEFORE the dropout
                    s[kt][q] = thr ? drop_apply(dc, xrow + (uint32_t)(kv0 + kt * 32 + acc_row(q, h)), p) : p;
                }
            rs += other_half(rs);
            if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                l_run *= alpha;
#pragma unroll
                for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                    for (int q = 0; q < 16; ++q) o[dt][q] *= alpha;
                m_run = m_new;
            }
            l_run += rs;
#pragma unroll
            for (int kt = 0; kt < C::NKT; ++kt)
#pragma unroll
                for (int st = 0; st < C::PV_STEPS; ++st) {
                    const u32x4 pf = pack_frag<P>(s[kt], st);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) P::mma(o[dt], v_frag<P>(vt_base, dt, kt, st, lane), pf);
                }
        }
        if (b + 1 < nb) stg.store(smem + (cur ^ 1) * C::STAGE, tid);
        __syncthreads();
    }
    const int qg = q0 + r;
    if (qg < Lq) {
        const float inv = 1.0f / l_run;
        if (h == 0) lse[(long)bh * Lp_q + qg] = m_run + log2f(l_run);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const long off = ((long)seq * Lq + qg) * ldo + head * 64 + dt * 32 + 8 * g + 4 * h;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = o[dt][4 * g + j] * inv;
                store_row4<P>(O, off, v[0], v[1], v[2], v[3]);
                if (P::IS_BF16 && O_lo) {      // what the 8-bit image dropped (tcdiff_attention_train: the backward's delta reads O + O_lo)
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] -= bf2f((uint16_t)(pack_bf2(v[j], 0.0f) & 0xffffu));
                    store_row4<P>(O_lo, off, v[0], v[1], v[2], v[3]);
                }
            }
    }
}

// =====================================================================================================================
// delta[bh][q] = sum_d dO[bh][q][d] * O[(seq Lq + q) ldo + head 64 + d]
// =====================================================================================================================
template <class P>
__global__ __launch_bounds__(256) void attn_delta_kernel(const typename P::elem_t* __restrict__ dO,
                                                         const typename P::elem_t* __restrict__ O,
                                                         const typename P::elem_t* __restrict__ O_lo, float* __restrict__ delta,
                                                         int n_bh, int H, int Lq, int Lp_q, int ldo) {
    // EPT = 16 bytes of a row per thread (bf16: 8 lanes per row, f32: 16): both rows are read as whole 128- / 256-byte lines
    // (a thread per row read 64 strided elements: 13 us per launch for 7 MB)
    typedef typename P::elem_t T;
    constexpr int EPT = 16 / sizeof(T), TPR = 64 / EPT;
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) / TPR;
    const int part = threadIdx.x % TPR;
    float s = 0.0f;
    const bool ok = i < (long)n_bh * Lq;
    int q = 0, bh = 0;
    if (ok) {
        q = (int)(i % Lq);
        bh = (int)(i / Lq);
        const int seq = bh / H, head = bh % H;
        const u32x4 av = *reinterpret_cast<const u32x4*>(dO + ((long)bh * Lp_q + q) * 64 + part * EPT);
        const u32x4 bv = *reinterpret_cast<const u32x4*>(O + ((long)seq * Lq + q) * ldo + head * 64 + part * EPT);
        if (P::IS_BF16) {
            u32x4 lv = {0u, 0u, 0u, 0u};
            if (O_lo) lv = *reinterpret_cast<const u32x4*>(O_lo + ((long)seq * Lq + q) * ldo + head * 64 + part * EPT);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s = fmaf(bf2f((uint16_t)(av[j] & 0xffffu)), bf2f((uint16_t)(bv[j] & 0xffffu)) + bf2f((uint16_t)(lv[j] & 0xffffu)), s);
                s = fmaf(bf2f((uint16_t)(av[j] >> 16)), bf2f((uint16_t)(bv[j] >> 16)) + bf2f((uint16_t)(lv[j] >> 16)), s);
            }
        } else {
            const f32x4_t af = __builtin_bit_cast(f32x4_t, av), bf = __builtin_bit_cast(f32x4_t, bv);
#pragma unroll
            for (int j = 0; j < 4; ++j) s = fmaf(af[j], bf[j], s);
        }
    }
#pragma unroll
    for (int m = 1; m < TPR; m <<= 1) s += __shfl_xor(s, m);
    if (ok && part == 0) delta[(long)bh * Lp_q + q] = s;
}

// =====================================================================================================================
// backward, query-major: dQ
// =====================================================================================================================
template <class P>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const char* __restrict__ Q, const char* __restrict__ K,
                                                          const char* __restrict__ V, const char* __restrict__ dO,
                                                          const float* __restrict__ lse, const float* __restrict__ delta,
                                                          void* __restrict__ dQ, int ld_dq, int H, int Lq, int Lk,
                                                          int Lp_q, int Lp_k, float scale_q, const int* __restrict__ seed,
                                                          int site, uint32_t thr, float dscale) {
    typedef AttnCfg<P> C;
    constexpr int ES = C::ES, KB = C::KB;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nqb = Lp_q / 128;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int qblk = (wg % nqb) * 128;
    const int head = (wg / nqb) % H, seq = wg / (nqb * H);
    if (qblk >= Lq) return;
    const int q0 = qblk + wave * 32;
    const bool active = q0 < Lq;
    const int bh = seq * H + head;
    const int qg = q0 + r;
    const DropCtx dc = drop_ctx(seed, site, thr, dscale);
    const uint32_t xrow = ((uint32_t)bh * (uint32_t)Lq + (uint32_t)qg) * (uint32_t)Lk;

    const char* Qg = Q + ((long)bh * Lp_q + qg) * 64 * ES;
    const char* Dg = dO + ((long)bh * Lp_q + qg) * 64 * ES;
    const char* Kg = K + (long)bh * Lp_k * 64 * ES;
    const char* Vg = V + (long)bh * Lp_k * 64 * ES;
    u32x4 qf[C::NKS], df[C::NKS];
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks) {
        qf[ks] = *reinterpret_cast<const u32x4*>(Qg + (2 * ks + h) * 16);
        df[ks] = *reinterpret_cast<const u32x4*>(Dg + (2 * ks + h) * 16);
    }
    // rows beyond Lq are padding: their lanes compute on zeros and are never stored
    const float my_lse = qg < Lq ? lse[(long)bh * Lp_q + qg] : 0.0f;
    const float my_delta = qg < Lq ? delta[(long)bh * Lp_q + qg] : 0.0f;

    TileStager<P> stg;
    f32x16_t dq[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int q = 0; q < 16; ++q) dq[dt][q] = 0.0f;

    const int nb = (Lk + KB - 1) / KB;
    stg.load(Kg, Vg, 0, tid);
    stg.store(smem, tid);
    __syncthreads();
    for (int b = 0; b < nb; ++b) {
        const int cur = b & 1, kv0 = b * KB;
        if (b + 1 < nb) stg.load(Kg, Vg, kv0 + KB, tid);
        const char* kt_base = smem + cur * C::STAGE;
        const char* vt_base = kt_base + C::TILE_BYTES;
        if (active) {
            constexpr float LOG2E = 1.4426950408889634f;
#pragma unroll
            for (int kt = 0; kt < C::NKT; ++kt) {
                f32x16_t s, dp;
#pragma unroll
                for (int q = 0; q < 16; ++q) { s[q] = 0.0f; dp[q] = 0.0f; }
                mma_tile_rows<P>(s, kt_base, kt * 32, qf, r, h);     // S^T  = K Q'^T
                mma_tile_rows<P>(dp, vt_base, kt * 32, df, r, h);    // dPd^T = V dO^T
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int key = kv0 + kt * 32 + acc_row(q, h);
                    const float p = key < Lk ? __builtin_amdgcn_exp2f(fmaf(s[q], LOG2E, -my_lse)) : 0.0f;
                    float g = dp[q];
                    if (thr) g = drop_apply(dc, xrow + (uint32_t)key, g);
                    s[q] = p * (g - my_delta);                       // dS^T
                }
#pragma unroll
                for (int st = 0; st < C::PV_STEPS; ++st) {
                    const u32x4 pf = pack_frag<P>(s, st);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) P::mma(dq[dt], v_frag<P>(kt_base, dt, kt, st, lane), pf);   // K^T dS^T
                }
            }
        }
        if (b + 1 < nb) stg.store(smem + (cur ^ 1) * C::STAGE, tid);
        __syncthreads();
    }
    if (qg < Lq) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                store_row4<P>(dQ, ((long)seq * Lq + qg) * ld_dq + head * 64 + dt * 32 + 8 * g + 4 * h,
                              dq[dt][4 * g + 0] * scale_q, dq[dt][4 * g + 1] * scale_q, dq[dt][4 * g + 2] * scale_q,
                              dq[dt][4 * g + 3] * scale_q);
    }
}

// =====================================================================================================================
// backward, key-major: dK, dV
// =====================================================================================================================
template <class P>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const char* __restrict__ Q, const char* __restrict__ K,
                                                           const char* __restrict__ V, const char* __restrict__ dO,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           void* __restrict__ dK, void* __restrict__ dV, int ld_dkv, int H,
                                                           int Lq, int Lk, int Lp_q, int Lp_k,
                                                           const int* __restrict__ seed, int site, uint32_t thr,
                                                           float dscale) {
    typedef AttnCfg<P> C;
    constexpr int ES = C::ES, KB = C::KB;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int nkb = Lp_k / 128;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int kblk = (wg % nkb) * 128;
    const int head = (wg / nkb) % H, seq = wg / (nkb * H);
    if (kblk >= Lk) return;
    const int k0 = kblk + wave * 32;
    const bool active = k0 < Lk;
    const int bh = seq * H + head;
    const int key = k0 + r;
    const bool key_ok = key < Lk;
    const DropCtx dc = drop_ctx(seed, site, thr, dscale);
    const uint32_t xbase = (uint32_t)bh * (uint32_t)Lq;

    const char* Kg = K + ((long)bh * Lp_k + key) * 64 * ES;
    const char* Vg = V + ((long)bh * Lp_k + key) * 64 * ES;
    const char* Qg = Q + (long)bh * Lp_q * 64 * ES;
    const char* Dg = dO + (long)bh * Lp_q * 64 * ES;
    const float* lse_b = lse + (long)bh * Lp_q;
    const float* del_b = delta + (long)bh * Lp_q;
    u32x4 kf[C::NKS], vf[C::NKS];
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks) {
        kf[ks] = *reinterpret_cast<const u32x4*>(Kg + (2 * ks + h) * 16);
        vf[ks] = *reinterpret_cast<const u32x4*>(Vg + (2 * ks + h) * 16);
    }
    TileStager<P> stg;
    f32x16_t dk[2], dv[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int q = 0; q < 16; ++q) { dk[dt][q] = 0.0f; dv[dt][q] = 0.0f; }

    const int nb = (Lq + KB - 1) / KB;           // query tiles (the Q' / dO images are zero beyond Lq, up to Lp_q)
    stg.load(Qg, Dg, 0, tid);
    stg.store(smem, tid);
    __syncthreads();
    for (int b = 0; b < nb; ++b) {
        const int cur = b & 1, q0 = b * KB;
        if (b + 1 < nb) stg.load(Qg, Dg, q0 + KB, tid);
        const char* qt_base = smem + cur * C::STAGE;
        const char* dt_base = qt_base + C::TILE_BYTES;
        if (active) {
            constexpr float LOG2E = 1.4426950408889634f;
#pragma unroll
            for (int qt = 0; qt < C::NKT; ++qt) {
                f32x16_t s, dp;
#pragma unroll
                for (int q = 0; q < 16; ++q) { s[q] = 0.0f; dp[q] = 0.0f; }
                mma_tile_rows<P>(s, qt_base, qt * 32, kf, r, h);     // S   = Q' K^T   (query in registers, key on the lane)
                mma_tile_rows<P>(dp, dt_base, qt * 32, vf, r, h);    // dPd = dO V^T
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int qi0 = q0 + qt * 32 + 8 * g + 4 * h;    // rows of registers 4g .. 4g+3 are consecutive queries
                    const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_b + qi0);
                    const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(del_b + qi0);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int q = 4 * g + t, qi = qi0 + t;
                        const bool ok = key_ok && qi < Lq;           // a select, not a product: pad rows may hold anything
                        const float p = ok ? __builtin_amdgcn_exp2f(fmaf(s[q], LOG2E, -l4[t])) : 0.0f;
                        const bool keep = thr ? drop_keep(dc, (xbase + (uint32_t)qi) * (uint32_t)Lk + (uint32_t)key) : true;
                        const float pd = keep ? p * dscale : 0.0f;
                        const float gd = keep ? dp[q] * dscale : 0.0f;
                        s[q] = pd;                                   // Pd
                        dp[q] = ok ? p * (gd - d4[t]) : 0.0f;        // dS
                    }
                }
#pragma unroll
                for (int st = 0; st < C::PV_STEPS; ++st) {
                    const u32x4 pf = pack_frag<P>(s, st), sf = pack_frag<P>(dp, st);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        P::mma(dv[dt], v_frag<P>(dt_base, dt, qt, st, lane), pf);    // dO^T Pd
                        P::mma(dk[dt], v_frag<P>(qt_base, dt, qt, st, lane), sf);    // Q'^T dS
                    }
                }
            }
        }
        if (b + 1 < nb) stg.store(smem + (cur ^ 1) * C::STAGE, tid);
        __syncthreads();
    }
    if (key_ok) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const long off = ((long)seq * Lk + key) * ld_dkv + head * 64 + dt * 32 + 8 * g + 4 * h;
                store_row4<P>(dK, off, dk[dt][4 * g + 0], dk[dt][4 * g + 1], dk[dt][4 * g + 2], dk[dt][4 * g + 3]);
                store_row4<P>(dV, off, dv[dt][4 * g + 0], dv[dt][4 * g + 1], dv[dt][4 * g + 2], dv[dt][4 * g + 3]);
            }
    }
}

// =====================================================================================================================
// backward, operand-resident variants (bf16, L <= 512): the structure of attn_res.h.  The streaming kernels above pay one
// barrier and one global -> register -> LDS hop per 64-row tile with 32 rows of independent work per wave (latency-bound:
// ~10 % of the MFMA rate at 450 x 450); here the two streamed operands of a (sequence, head) are loaded ONCE by LDS-DMA
// (2 x <= 64 KB), a wave owns NG groups of 32 rows that share every fragment read, and the loop has no barrier.
//   dq:  K, V resident; a wave owns 32 NG queries (Q', dO fragments, lse, delta in registers)
//   dkv: Q', dO resident (+ lse, delta rows); a wave owns 32 NG keys (K, V fragments in registers)
// =====================================================================================================================
DEVINL void res_stage_images(char* A_s, char* B_s, const char* Ag, const char* Bg, int nt, int wave, int lane) {
    for (int blk = wave; blk < nt * 8; blk += 8) {      // blk = 8 consecutive rows
        const int row = blk * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ tile_swz(row);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(Ag + (long)row * 128 + chunk * 16), (lds_void_t*)(A_s + blk * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(Bg + (long)row * 128 + chunk * 16), (lds_void_t*)(B_s + blk * 1024), 16, 0, 0);
    }
}

// 32 output rows x 64 features of one group through 4 KB of wave-private LDS -> 16-byte pieces of full 128-byte rows
// (attn_res.h's output path): acc[dt] holds features dt*32 + 8 g + 4 h + {0..3} of row r
DEVINL void res_store_rows(char* stg, const f32x16_t (&acc)[2], float scale, uint16_t* dst, long row0_elem, int ld, int rows_ok,
                           int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            uint2 pk;
            pk.x = pack_bf2(acc[dt][4 * q4 + 0] * scale, acc[dt][4 * q4 + 1] * scale);
            pk.y = pack_bf2(acc[dt][4 * q4 + 2] * scale, acc[dt][4 * q4 + 3] * scale);
            *reinterpret_cast<uint2*>(stg + r * 128 + (((4 * dt + q4) ^ ((r >> 1) & 7)) << 4) + 8 * h) = pk;
        }
    const int srow0 = lane >> 3, sch = lane & 7;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int row = srow0 + 8 * k;
        const u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * 128 + ((sch ^ ((row >> 1) & 7)) << 4));
        if (row < rows_ok) *reinterpret_cast<u32x4*>(dst + row0_elem + (long)row * ld + sch * 8) = v;
    }
}

template <int NG>
__global__ __launch_bounds__(512) void attn_bwd_dq_res_kernel(const char* __restrict__ Q, const char* __restrict__ K,
                                                              const char* __restrict__ V, const char* __restrict__ dO,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              uint16_t* __restrict__ dQ, int ld_dq, int H, int Lq, int Lk,
                                                              int Lp_q, int Lp_k, float scale_q, const int* __restrict__ seed,
                                                              int site, uint32_t thr, float dscale) {
    typedef MmaBF16 P;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int nqb = (Lq + 256 * NG - 1) / (256 * NG);
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int qb = wg % nqb, head = (wg / nqb) % H, seq = wg / (nqb * H);
    const int bh = seq * H + head;
    const int nt = (Lk + 63) / 64;                       // <= 8 (the launcher checks Lk <= 512)
    char* Ks = smem;
    char* Vs = smem + nt * 8192;
    res_stage_images(Ks, Vs, K + (long)bh * Lp_k * 128, V + (long)bh * Lp_k * 128, nt, wave, lane);
    const int qbase = qb * 256 * NG + wave * 32 * NG;
    u32x4 qf[NG][4], df[NG][4];
    float my_lse[NG], my_delta[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int qg = qbase + g * 32 + r;
        const int qrow = qg < Lp_q ? qg : Lp_q - 1;
        const char* Qg = Q + ((long)bh * Lp_q + qrow) * 128;
        const char* Dg = dO + ((long)bh * Lp_q + qrow) * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[g][ks] = *reinterpret_cast<const u32x4*>(Qg + (2 * ks + h) * 16);
            df[g][ks] = *reinterpret_cast<const u32x4*>(Dg + (2 * ks + h) * 16);
        }
        my_lse[g] = qg < Lq ? lse[(long)bh * Lp_q + qg] : 0.0f;
        my_delta[g] = qg < Lq ? delta[(long)bh * Lp_q + qg] : 0.0f;
    }
    const DropCtx dc = drop_ctx(seed, site, thr, dscale);
    f32x16_t dq[NG][2];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int q = 0; q < 16; ++q) dq[g][dt][q] = 0.0f;
    sync_dma();
    const int ngrp = qbase >= Lq ? 0 : (NG > 1 && qbase + 32 < Lq ? NG : 1);      // wave-uniform: active row groups
    if (ngrp > 0) {
        constexpr float LOG2E = 1.4426950408889634f;
        const int nkt = (Lk + 31) / 32;
#pragma unroll 1
        for (int kt = 0; kt < nkt; ++kt) {
            const char* kt_base = Ks + (kt >> 1) * 8192;
            const char* vt_base = Vs + (kt >> 1) * 8192;
            const int k32 = kt & 1, kv0 = kt * 32;
            f32x16_t s[NG], dp[NG];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(kt_base + tile_off(k32 * 32 + r, 2 * ks + h));
                const u32x4 vf = *reinterpret_cast<const u32x4*>(vt_base + tile_off(k32 * 32 + r, 2 * ks + h));
                if (ks == 0) {
                    const f32x16_t z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int g = 0; g < NG; ++g) { s[g] = z; dp[g] = z; }
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    P::mma(s[g], kf, qf[g][ks]);         // S^T   = K Q'^T
                    P::mma(dp[g], vf, df[g][ks]);        // dPd^T = V dO^T
                }
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const uint32_t xrow = ((uint32_t)bh * (uint32_t)Lq + (uint32_t)(qbase + g * 32 + r)) * (uint32_t)Lk;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int key = kv0 + acc_row(q, h);
                    const float p = key < Lk ? __builtin_amdgcn_exp2f(fmaf(s[g][q], LOG2E, -my_lse[g])) : 0.0f;
                    float gd = dp[g][q];
                    if (thr) gd = drop_apply(dc, xrow + (uint32_t)key, gd);
                    s[g][q] = p * (gd - my_delta[g]);    // dS^T
                }
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                u32x4 pf[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) pf[g] = pack_frag<P>(s[g], st);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const u32x4 ktf = v_frag<P>(kt_base, dt, k32, st, lane);
#pragma unroll
                    for (int g = 0; g < NG; ++g) P::mma(dq[g][dt], ktf, pf[g]);      // dQ'^T += K^T dS^T
                }
            }
        }
    }
    char* stg = smem + 2 * nt * 8192 + wave * 4096;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g >= ngrp) continue;
        const int q0 = qbase + g * 32;
        res_store_rows(stg, dq[g], scale_q, dQ, ((long)seq * Lq + q0) * ld_dq + head * 64, ld_dq, Lq - q0, lane);
    }
}

template <int NG>
__global__ __launch_bounds__(512) void attn_bwd_dkv_res_kernel(const char* __restrict__ Q, const char* __restrict__ K,
                                                               const char* __restrict__ V, const char* __restrict__ dO,
                                                               const float* __restrict__ lse, const float* __restrict__ delta,
                                                               uint16_t* __restrict__ dK, uint16_t* __restrict__ dV, int ld_dkv,
                                                               int H, int Lq, int Lk, int Lp_q, int Lp_k,
                                                               const int* __restrict__ seed, int site, uint32_t thr,
                                                               float dscale) {
    typedef MmaBF16 P;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int nkb = (Lk + 256 * NG - 1) / (256 * NG);
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int kb = wg % nkb, head = (wg / nkb) % H, seq = wg / (nkb * H);
    const int bh = seq * H + head;
    const int nt = (Lq + 63) / 64;                       // query tiles, <= 8
    char* Qs = smem;
    char* Ds = smem + nt * 8192;
    float* lse_s = reinterpret_cast<float*>(smem + 2 * nt * 8192);      // [nt * 64] each: lse, then delta
    float* del_s = lse_s + nt * 64;
    res_stage_images(Qs, Ds, Q + (long)bh * Lp_q * 128, dO + (long)bh * Lp_q * 128, nt, wave, lane);
    for (int i = tid; i < nt * 64; i += 512) {
        const bool ok = i < Lq;                          // rows beyond Lq: p is forced to 0 below, any finite value does
        lse_s[i] = ok ? lse[(long)bh * Lp_q + i] : 0.0f;
        del_s[i] = ok ? delta[(long)bh * Lp_q + i] : 0.0f;
    }
    const int kbase = kb * 256 * NG + wave * 32 * NG;
    u32x4 kf[NG][4], vf[NG][4];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int key = kbase + g * 32 + r;
        const int krow = key < Lp_k ? key : Lp_k - 1;
        const char* Kg = K + ((long)bh * Lp_k + krow) * 128;
        const char* Vg = V + ((long)bh * Lp_k + krow) * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[g][ks] = *reinterpret_cast<const u32x4*>(Kg + (2 * ks + h) * 16);
            vf[g][ks] = *reinterpret_cast<const u32x4*>(Vg + (2 * ks + h) * 16);
        }
    }
    const DropCtx dc = drop_ctx(seed, site, thr, dscale);
    f32x16_t dk[NG][2], dv[NG][2];
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int q = 0; q < 16; ++q) { dk[g][dt][q] = 0.0f; dv[g][dt][q] = 0.0f; }
    sync_dma();
    const int ngrp = kbase >= Lk ? 0 : (NG > 1 && kbase + 32 < Lk ? NG : 1);
    if (ngrp > 0) {
        constexpr float LOG2E = 1.4426950408889634f;
        const int nqt = (Lq + 31) / 32;
#pragma unroll 1
        for (int qt = 0; qt < nqt; ++qt) {
            const char* qt_base = Qs + (qt >> 1) * 8192;
            const char* dt_base = Ds + (qt >> 1) * 8192;
            const int q32 = qt & 1, q0 = qt * 32;
            f32x16_t s[NG], dp[NG];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const u32x4 qfr = *reinterpret_cast<const u32x4*>(qt_base + tile_off(q32 * 32 + r, 2 * ks + h));
                const u32x4 dfr = *reinterpret_cast<const u32x4*>(dt_base + tile_off(q32 * 32 + r, 2 * ks + h));
                if (ks == 0) {
                    const f32x16_t z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                    for (int g = 0; g < NG; ++g) { s[g] = z; dp[g] = z; }
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    P::mma(s[g], qfr, kf[g][ks]);        // S   = Q' K^T  (query in registers, key on the lane)
                    P::mma(dp[g], dfr, vf[g][ks]);       // dPd = dO V^T
                }
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int qi0 = q0 + 8 * g4 + 4 * h;     // registers 4 g4 .. 4 g4 + 3 are consecutive queries
                const f32x4_t l4 = *reinterpret_cast<const f32x4_t*>(lse_s + qi0);
                const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(del_s + qi0);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const int key = kbase + g * 32 + r;
                    const bool key_ok = key < Lk;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int q = 4 * g4 + t, qi = qi0 + t;
                        const bool ok = key_ok && qi < Lq;
                        const float p = ok ? __builtin_amdgcn_exp2f(fmaf(s[g][q], LOG2E, -l4[t])) : 0.0f;
                        const bool keep = thr ? drop_keep(dc, ((uint32_t)bh * (uint32_t)Lq + (uint32_t)qi) * (uint32_t)Lk + (uint32_t)key) : true;
                        const float pd = keep ? p * dscale : 0.0f;
                        const float gd = keep ? dp[g][q] * dscale : 0.0f;
                        s[g][q] = pd;                                    // Pd
                        dp[g][q] = ok ? p * (gd - d4[t]) : 0.0f;         // dS
                    }
                }
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                u32x4 pf[NG], sf[NG];
#pragma unroll
                for (int g = 0; g < NG; ++g) { pf[g] = pack_frag<P>(s[g], st); sf[g] = pack_frag<P>(dp[g], st); }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const u32x4 dOt = v_frag<P>(dt_base, dt, q32, st, lane);
                    const u32x4 Qt = v_frag<P>(qt_base, dt, q32, st, lane);
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        P::mma(dv[g][dt], dOt, pf[g]);   // dV^T += dO^T Pd
                        P::mma(dk[g][dt], Qt, sf[g]);    // dK^T += Q'^T dS
                    }
                }
            }
        }
    }
    // the output staging areas take the place of the Q' image: every wave must be out of the loop first
    __syncthreads();
    char* stg = smem + wave * 4096;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g >= ngrp) continue;
        const int k0 = kbase + g * 32;
        const long off = ((long)seq * Lk + k0) * ld_dkv + head * 64;
        res_store_rows(stg, dk[g], 1.0f, dK, off, ld_dkv, Lk - k0, lane);
        res_store_rows(stg, dv[g], 1.0f, dV, off, ld_dkv, Lk - k0, lane);
    }
}

// =====================================================================================================================
// C ABI
// =====================================================================================================================
static bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int tcdiff_attention_train(int dtype, const void* Q, const void* K, const void* V, void* O, void* O_lo, float* lse,
                                      int n_seq, int H, int Lq, int Lk, int Lp_q, int Lp_k, int ldo, const int* seed,
                                      int site, uint32_t drop_thr, float drop_scale, hipStream_t stream) {
    if (!Q || !K || !V || !O || !lse || n_seq <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return TC_ERR_ARG;
    if (O_lo && (dtype != TC_DTYPE_BF16 || !a16(O_lo))) return TC_ERR_ARG;
    if (dtype != TC_DTYPE_BF16 && dtype != TC_DTYPE_F32) return TC_ERR_ARG;
    if (Lp_q % 128 != 0 || Lp_k % 64 != 0 || Lp_q < Lq || Lp_k < Lk || ldo < H * 64 || ldo % 4 != 0) return TC_ERR_ARG;
    if (!a16(Q) || !a16(K) || !a16(V) || !a16(O)) return TC_ERR_ALIGN;
    if (dtype == TC_DTYPE_BF16 && Lp_q >= 512 && ldo % 8 == 0) {
        // the K/V-resident kernel of the sampler (attn_res.h) with dropout and lse: 450 x 450 at 32 sequences 61 -> 3x us
        const int ntm = (Lk + 63) / 64 < ATT_RES_MAXT ? (Lk + 63) / 64 : ATT_RES_MAXT;
        const int smem_bytes = 2 * ntm * 8192 + 8 * 4096;
        static tc_dev_state dev_state;
        const int n_cu = tc_device_once(dev_state, [](int) {
            hipError_t a = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_res_kernel<1, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ATT_RES_MAXT * 8192 + 8 * 4096);
            hipError_t b = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_res_kernel<2, true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ATT_RES_MAXT * 8192 + 8 * 4096);
            return a != hipSuccess ? a : b;
        });
        if (n_cu < 0) return n_cu;
        const int ng = ((Lq + 255) / 256) * H * n_seq <= n_cu ? 1 : 2;
        const int nqb = (Lq + 256 * ng - 1) / (256 * ng);
        const AttnTrainArgs ta = {lse, seed, site, drop_thr, drop_scale, (char*)O_lo};
        if (ng == 1)
            hipLaunchKernelGGL((attention_res_kernel<1, true>), dim3(nqb * H * n_seq), dim3(512), smem_bytes, stream,
                               (const char*)Q, (const char*)K, (const char*)V, (char*)O, H, Lq, Lk, Lp_q, Lp_k, ldo, 0, ta);
        else
            hipLaunchKernelGGL((attention_res_kernel<2, true>), dim3(nqb * H * n_seq), dim3(512), smem_bytes, stream,
                               (const char*)Q, (const char*)K, (const char*)V, (char*)O, H, Lq, Lk, Lp_q, Lp_k, ldo, 0, ta);
        TC_CHECK_LAUNCH();
        return TC_OK;
    }
    dim3 grid((Lp_q / 128) * H * n_seq);
    if (dtype == TC_DTYPE_BF16)
        hipLaunchKernelGGL(attention_train_kernel<MmaBF16>, grid, dim3(256), 0, stream, (const char*)Q, (const char*)K,
                           (const char*)V, (char*)O, (char*)O_lo, lse, H, Lq, Lk, Lp_q, Lp_k, ldo, seed, site, drop_thr, drop_scale);
    else
        hipLaunchKernelGGL(attention_train_kernel<MmaF32>, grid, dim3(256), 0, stream, (const char*)Q, (const char*)K,
                           (const char*)V, (char*)O, (char*)nullptr, lse, H, Lq, Lk, Lp_q, Lp_k, ldo, seed, site, drop_thr, drop_scale);
    TC_CHECK_LAUNCH();
    return TC_OK;
}

template <class P>
static void launch_attn_bwd(const void* Q, const void* K, const void* V, const void* O, const void* O_lo, const void* dO, const float* lse,
                            float* delta, void* dQ, int ld_dq, void* dK, void* dV, int ld_dkv, int n_seq, int H, int Lq,
                            int Lk, int Lp_q, int Lp_k, int ldo, float scale_q, const int* seed, int site, uint32_t thr,
                            float dscale, hipStream_t stream) {
    typedef typename P::elem_t T;
    const long nd = (long)n_seq * H * Lq;
    const long ndt = nd * (64 / (16 / (long)sizeof(T)));      // threads: 16 bytes of a row each
    hipLaunchKernelGGL(attn_delta_kernel<P>, dim3((unsigned)((ndt + 255) / 256)), dim3(256), 0, stream, (const T*)dO,
                       (const T*)O, (const T*)O_lo, delta, n_seq * H, H, Lq, Lp_q, ldo);
    hipLaunchKernelGGL(attn_bwd_dq_kernel<P>, dim3((Lp_q / 128) * H * n_seq), dim3(256), 0, stream, (const char*)Q,
                       (const char*)K, (const char*)V, (const char*)dO, lse, delta, dQ, ld_dq, H, Lq, Lk, Lp_q, Lp_k, scale_q,
                       seed, site, thr, dscale);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel<P>, dim3((Lp_k / 128) * H * n_seq), dim3(256), 0, stream, (const char*)Q,
                       (const char*)K, (const char*)V, (const char*)dO, lse, delta, dK, dV, ld_dkv, H, Lq, Lk, Lp_q, Lp_k, seed,
                       site, thr, dscale);
}

extern "C" int tcdiff_attention_bwd(int dtype, const void* Q, const void* K, const void* V, const void* O, const void* O_lo,
                                    const void* dO,
                                    const float* lse, float* delta, void* dQ, int ld_dq, void* dK, void* dV, int ld_dkv,
                                    int n_seq, int H, int Lq, int Lk, int Lp_q, int Lp_k, int ldo, float scale_q,
                                    const int* seed, int site, uint32_t drop_thr, float drop_scale, hipStream_t stream) {
    if (!Q || !K || !V || !O || !dO || !lse || !delta || !dQ || !dK || !dV || n_seq <= 0 || H <= 0 || Lq <= 0 || Lk <= 0)
        return TC_ERR_ARG;
    if (dtype != TC_DTYPE_BF16 && dtype != TC_DTYPE_F32) return TC_ERR_ARG;
    if (O_lo && (dtype != TC_DTYPE_BF16 || !a16(O_lo))) return TC_ERR_ARG;
    if (Lp_q % 128 != 0 || Lp_k % 128 != 0 || Lp_q < Lq || Lp_k < Lk || ldo < H * 64 || ld_dq < H * 64 || ld_dkv < H * 64 ||
        ld_dq % 4 != 0 || ld_dkv % 4 != 0)
        return TC_ERR_ARG;
    if (!a16(Q) || !a16(K) || !a16(V) || !a16(O) || !a16(dO) || !a16(dQ) || !a16(dK) || !a16(dV) || !a16(lse) || !a16(delta) ||
        ((long)ldo * (dtype == TC_DTYPE_BF16 ? 2 : 4)) % 16 != 0)         // delta reads 16-byte pieces of O's rows
        return TC_ERR_ALIGN;
    if (dtype == TC_DTYPE_BF16 && Lq >= 256 && Lq <= 512 && Lk <= 512 && ld_dq % 8 == 0 && ld_dkv % 8 == 0) {
        // operand-resident kernels (the training shapes: 450 x 450 self-attention, 450 x 152 cross-attention)
        constexpr int DQ_NG = 2, DKV_NG = TC_DKV_NG;
        const int ntk = (Lk + 63) / 64, ntq = (Lq + 63) / 64;
        const int smem_dq = 2 * ntk * 8192 + 8 * 4096;
        const int dkv_need = 2 * ntq * 8192 + 2 * ntq * 64 * 4;
        const int smem_dkv = dkv_need > 8 * 4096 ? dkv_need : 8 * 4096;
        static tc_dev_state dev_state;
        const int n_cu = tc_device_once(dev_state, [](int) {
            hipError_t a = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_res_kernel<DQ_NG>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 8192 + 8 * 4096);
            hipError_t b = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_res_kernel<DKV_NG>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 8 * 8192 + 2 * 8 * 64 * 4);
            return a != hipSuccess ? a : b;
        });
        if (n_cu < 0) return n_cu;
        const long nd = (long)n_seq * H * Lq;
        hipLaunchKernelGGL(attn_delta_kernel<MmaBF16>, dim3((unsigned)((nd * 8 + 255) / 256)), dim3(256), 0, stream,
                           (const uint16_t*)dO, (const uint16_t*)O, (const uint16_t*)O_lo, delta, n_seq * H, H, Lq, Lp_q, ldo);
        const int nqb = (Lq + 256 * DQ_NG - 1) / (256 * DQ_NG), nkb = (Lk + 256 * DKV_NG - 1) / (256 * DKV_NG);
        hipLaunchKernelGGL(attn_bwd_dq_res_kernel<DQ_NG>, dim3(nqb * H * n_seq), dim3(512), smem_dq, stream, (const char*)Q,
                           (const char*)K, (const char*)V, (const char*)dO, lse, delta, (uint16_t*)dQ, ld_dq, H, Lq, Lk, Lp_q,
                           Lp_k, scale_q, seed, site, drop_thr, drop_scale);
        hipLaunchKernelGGL(attn_bwd_dkv_res_kernel<DKV_NG>, dim3(nkb * H * n_seq), dim3(512), smem_dkv, stream, (const char*)Q,
                           (const char*)K, (const char*)V, (const char*)dO, lse, delta, (uint16_t*)dK, (uint16_t*)dV, ld_dkv, H,
                           Lq, Lk, Lp_q, Lp_k, seed, site, drop_thr, drop_scale);
    } else if (dtype == TC_DTYPE_BF16)
        launch_attn_bwd<MmaBF16>(Q, K, V, O, O_lo, dO, lse, delta, dQ, ld_dq, dK, dV, ld_dkv, n_seq, H, Lq, Lk, Lp_q, Lp_k, ldo,
                                 scale_q, seed, site, drop_thr, drop_scale, stream);
    else
        launch_attn_bwd<MmaF32>(Q, K, V, O, nullptr, dO, lse, delta, dQ, ld_dq, dK, dV, ld_dkv, n_seq, H, Lq, Lk, Lp_q, Lp_k, ldo,
                                scale_q, seed, site, drop_thr, drop_scale, stream);
    TC_CHECK_LAUNCH();
    return TC_OK;
}
