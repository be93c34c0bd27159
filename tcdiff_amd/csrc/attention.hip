// Fused softmax(Q K^T) V for the TCDiff denoiser (d_k = 64 per head), gfx950.
//
// Replaces the materialised-score attention of SBI_MSA (model/model.py:97-102: matmul, softmax over the
// flattened frame x dancer token axis, matmul) and of the music encoder's nn.MultiheadAttention
// (model/model.py:228-236).  One joint softmax over all L = frames*dancers keys, as in the reference.
//
// Structure (per workgroup = 4 waves = 128 query rows of one (sequence, head)):
//   * "swapped" products: S^T = K Q^T and O^T = V^T P^T, so the query index sits on the MFMA lane and the
//     key / feature index in the accumulator registers.  The row max / row sum of the softmax are then
//     in-register reductions plus ONE cross-half shuffle, and the P^T accumulator tile is already the B
//     operand of the PV MFMA (cdna_hip_programming.md section 3 "accumulator tile as the next operand"):
//     no LDS round trip for P.
//   * K and V tiles [KB keys][64] stream through LDS in their natural [key][d] layout (double buffered,
//     XOR-swizzled 16-byte chunks), KB = 64 keys (bf16) / 32 keys (f32); online softmax across tiles (fp32 m, l).
//   * the PV A operand is V^T: bf16 reads it with the hardware transpose read ds_read_b64_tr_b16 (two reads per
//     fragment, in the key order the P^T registers are in); f32 reads one dword per MFMA (lanes = consecutive d).
#include "common.h"
#include <type_traits>
#include "tcdiff_hip.h"

#include "attn_common.h"

template <class P>
__global__ __launch_bounds__(256) void attention_kernel(const char* __restrict__ Q, const char* __restrict__ K,
                                                        const char* __restrict__ V, char* __restrict__ O, int H,
                                                        int Lq, int Lk, int Lp_q, int Lp_k, int ldo,
                                                        int n_shared) {
    typedef AttnCfg<P> C;
    typedef typename P::elem_t T;
    constexpr int ES = C::ES, KB = C::KB;
    __shared__ __attribute__((aligned(16))) char smem[2 * C::STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    // 1-D grid, XCD-remapped: the query blocks of one (sequence, head) share K/V through one XCD's L2, and an XCD
    // owns a contiguous range of sequences (the same rows it owns in the GEMMs)
    const int nqb = Lp_q / 128;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int qblk = (wg % nqb) * 128;
    const int head = (wg / nqb) % H, seq = wg / (nqb * H);
    if (qblk >= Lq) return;
    const int q0 = qblk + wave * 32;
    const bool active = q0 < Lq;   // wave-uniform
    const int kv = seq < n_shared ? 0 : seq - n_shared + (n_shared > 0 ? 1 : 0);

    const char* Qg = Q + ((long)(seq * H + head) * Lp_q + q0 + r) * 64 * ES;
    const char* Kg = K + (long)(kv * H + head) * Lp_k * 64 * ES;
    const char* Vg = V + (long)(kv * H + head) * Lp_k * 64 * ES;

    // Q^T fragments stay in registers for the whole kernel
    u32x4 qf[C::NKS];
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks) qf[ks] = *reinterpret_cast<const u32x4*>(Qg + (2 * ks + h) * 16);

    // staging: 512 16-byte chunks per tile, 2 per thread per tile; a tile is KB consecutive rows = 8 KB contiguous
    u32x4 sk[2], sv[2];
    auto load_tiles = [&](int kv0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * 256;
            sk[i] = *reinterpret_cast<const u32x4*>(Kg + (long)kv0 * 64 * ES + c * 16);
            sv[i] = *reinterpret_cast<const u32x4*>(Vg + (long)kv0 * 64 * ES + c * 16);
        }
    };
    auto store_tiles = [&](char* stage) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int c = tid + i * 256;
            const int row = c / (8 * C::DSUB), chk = c % (8 * C::DSUB);
            const int off = (chk >> 3) * (KB * TC_ROWB) + tile_off(row, chk & 7);
            *reinterpret_cast<u32x4*>(stage + off) = sk[i];
            *reinterpret_cast<u32x4*>(stage + C::TILE_BYTES + off) = sv[i];
        }
    };

    f32x16_t o[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int q = 0; q < 16; ++q) o[dt][q] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    const int nb = (Lk + KB - 1) / KB;
    load_tiles(0);
    store_tiles(smem);
    __syncthreads();

    for (int b = 0; b < nb; ++b) {
        const int cur = b & 1;
        const int kv0 = b * KB;
        if (b + 1 < nb) load_tiles(kv0 + KB);
        const char* kt_base = smem + cur * C::STAGE;
        const char* vt_base = kt_base + C::TILE_BYTES;

        if (active) {   // waves whose 32 query rows are all padding only help with the staging
        // ---- S^T = K Q^T ---------------------------------------------------------------------------
        f32x16_t s[C::NKT];
#pragma unroll
        for (int kt = 0; kt < C::NKT; ++kt) {
#pragma unroll
            for (int q = 0; q < 16; ++q) s[kt][q] = 0.0f;
            if constexpr (P::IS_X3) {     // split-bf16: the d k-steps in pairs (common.h MmaBF16x3::mma2)
#pragma unroll
                for (int ks = 0; ks < C::NKS; ks += 2) {
                    const int sub = ks >> 2, ch = 2 * (ks & 3) + h;
                    const char* kp = kt_base + sub * (KB * TC_ROWB);
                    const u32x4 k0 = *reinterpret_cast<const u32x4*>(kp + tile_off(kt * 32 + r, ch));
                    const u32x4 k1 = *reinterpret_cast<const u32x4*>(kp + tile_off(kt * 32 + r, ch + 2));
                    P::mma2(s[kt], k0, k1, qf[ks], qf[ks + 1]);
                }
            } else {
#pragma unroll
            for (int ks = 0; ks < C::NKS; ++ks) {
                const int sub = ks >> 2, ch = 2 * (ks & 3) + h;
                u32x4 kf = *reinterpret_cast<const u32x4*>(kt_base + sub * (KB * TC_ROWB) + tile_off(kt * 32 + r, ch));
                P::mma(s[kt], kf, qf[ks]);
            }
            }
        }
        // ---- mask keys beyond Lk (last tile only) -------------------------------------------------
        if (kv0 + KB > Lk) {
#pragma unroll
            for (int kt = 0; kt < C::NKT; ++kt)
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (kv0 + kt * 32 + acc_row(q, h) >= Lk) s[kt][q] = -INFINITY;
        }
        // ---- online softmax in base 2: p = 2^(s*log2e - m), one fma + one v_exp_f32 per score.  This lane owns
        //      query q0 + r and half of the keys; the other half sits 32 lanes away.
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
                s[kt][q] = p;
                rs += p;
            }
        rs += other_half(rs);
        if (__builtin_amdgcn_ballot_w64(m_new > m_run) != 0) {   // some row's running max grew: rescale (wave-uniform)
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int q = 0; q < 16; ++q) o[dt][q] *= alpha;
            m_run = m_new;
        }
        l_run += rs;

        // ---- O^T += V^T P^T --------------------------------------------------------------------------
#pragma unroll
        for (int kt = 0; kt < C::NKT; ++kt) {
            if constexpr (P::IS_X3) {     // split-bf16: the 8-key k-steps of the tile in pairs
#pragma unroll
                for (int st = 0; st < C::PV_STEPS; st += 2) {
                    const f32x4_t p0 = {s[kt][4 * st + 0], s[kt][4 * st + 1], s[kt][4 * st + 2], s[kt][4 * st + 3]};
                    const f32x4_t p1 = {s[kt][4 * st + 4], s[kt][4 * st + 5], s[kt][4 * st + 6], s[kt][4 * st + 7]};
                    const u32x4 pf0 = P::chunk_from4(p0), pf1 = P::chunk_from4(p1);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const u32x4 v0 = v_frag<P>(vt_base, dt, kt, st, lane), v1 = v_frag<P>(vt_base, dt, kt, st + 1, lane);
                        P::mma2(o[dt], v0, v1, pf0, pf1);
                    }
                }
                continue;
            }
#pragma unroll
            for (int st = 0; st < C::PV_STEPS; ++st) {
                u32x4 pf;
                if (P::IS_BF16) {
                    // registers 8st..8st+7 hold keys 16st + 8(j>>2) + 4h + (j&3), j = 0..7
                    pf[0] = pack_bf2(s[kt][8 * st + 0], s[kt][8 * st + 1]);
                    pf[1] = pack_bf2(s[kt][8 * st + 2], s[kt][8 * st + 3]);
                    pf[2] = pack_bf2(s[kt][8 * st + 4], s[kt][8 * st + 5]);
                    pf[3] = pack_bf2(s[kt][8 * st + 6], s[kt][8 * st + 7]);
                } else {
                    // f32: MFMA j of the k-step pairs register 4st+j of both halves: keys 8st + j and 8st + 4 + j
                    const f32x4_t pv = {s[kt][4 * st + 0], s[kt][4 * st + 1], s[kt][4 * st + 2], s[kt][4 * st + 3]};
                    pf = P::chunk_from4(pv);  // f32: whole-vector cast (element-wise bit_cast is miscompiled); bf16x3: the (hi, lo) split
                }
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    const u32x4 vf = v_frag<P>(vt_base, dt, kt, st, lane);
                    P::mma(o[dt], vf, pf);
                }
            }
        }
        }  // active
        if (b + 1 < nb) store_tiles(smem + (cur ^ 1) * C::STAGE);
        __syncthreads();
    }

    // ---- O[q][d] = O^T[d][q] / l ---------------------------------------------------------------------
    const int qg = q0 + r;
    if (qg < Lq) {
        const float inv = 1.0f / l_run;
        T* orow = reinterpret_cast<T*>(O) + ((long)seq * Lq + qg) * ldo + head * 64;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d0 = dt * 32 + 8 * g + 4 * h;
                float v0 = o[dt][4 * g + 0] * inv, v1 = o[dt][4 * g + 1] * inv;
                float v2 = o[dt][4 * g + 2] * inv, v3 = o[dt][4 * g + 3] * inv;
                if (P::IS_BF16) {
                    uint2 pk;
                    pk.x = pack_bf2(v0, v1);
                    pk.y = pack_bf2(v2, v3);
                    *reinterpret_cast<uint2*>(orow + d0) = pk;
                } else {
                    const f32x4_t pk = {v0, v1, v2, v3};
                    *reinterpret_cast<u32x4*>(orow + d0) = P::chunk_from4(pk);
                }
            }
    }
}


#include "attn_res.h"       // attention_res_kernel: the K/V-resident variant (shared with attention_train.hip)

extern "C" int tcdiff_attention(int dtype, const void* Q, const void* K, const void* V, void* O, int n_seq, int H,
                                int Lq, int Lk, int Lp_q, int Lp_k, int ldo, int n_shared, int ng, hipStream_t stream) {
    if (!Q || !K || !V || !O || n_seq <= 0 || H <= 0 || Lq <= 0 || Lk <= 0) return TC_ERR_ARG;
    if (dtype != TC_DTYPE_BF16 && dtype != TC_DTYPE_F32 && dtype != TC_DTYPE_BF16X3) return TC_ERR_ARG;
    if (ng < 0 || ng > 2) return TC_ERR_ARG;
    if (Lp_q % 128 != 0 || Lp_k % 64 != 0 || Lp_q < Lq || Lp_k < Lk || ldo < H * 64 || ldo % 4 != 0) return TC_ERR_ARG;
    if (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)O) & 15) return TC_ERR_ALIGN;
    if (dtype == TC_DTYPE_BF16 && Lp_q >= 512) {
        if (ldo % 8 != 0) return TC_ERR_ALIGN;      // output rows leave as 16-byte pieces
        // K/V-resident kernel (Q image padded to >= 512 rows); keys beyond 512 come in LDS-resident chunks
        const int ntm = (Lk + 63) / 64 < ATT_RES_MAXT ? (Lk + 63) / 64 : ATT_RES_MAXT;
        const int smem_bytes = 2 * ntm * 8192 + 8 * 4096;      // K, V images + the eight 4-KB output staging areas
        // 32 query rows per wave (two workgroups per 450-token sequence) while that still fits one round over the CUs,
        // 64 rows per wave (K / V fragment reads shared by two row groups) beyond
        static tc_dev_state dev_state;
        const int n_cu = tc_device_once(dev_state, [](int) {
            hipError_t a = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_res_kernel<1, false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ATT_RES_MAXT * 8192 + 8 * 4096);
            hipError_t b = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_res_kernel<2, false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * ATT_RES_MAXT * 8192 + 8 * 4096);
            return a != hipSuccess ? a : b;
        });
        if (n_cu < 0) return n_cu;
        if (ng == 0) ng = ((Lq + 255) / 256) * H * n_seq <= n_cu ? 1 : 2;
        const int nqb = (Lq + 256 * ng - 1) / (256 * ng);
        if (ng == 1)
            hipLaunchKernelGGL((attention_res_kernel<1, false>), dim3(nqb * H * n_seq), dim3(512), smem_bytes, stream,
                               (const char*)Q, (const char*)K, (const char*)V, (char*)O, H, Lq, Lk, Lp_q, Lp_k, ldo, n_shared,
                               AttnTrainArgs{});
        else
            hipLaunchKernelGGL((attention_res_kernel<2, false>), dim3(nqb * H * n_seq), dim3(512), smem_bytes, stream,
                               (const char*)Q, (const char*)K, (const char*)V, (char*)O, H, Lq, Lk, Lp_q, Lp_k, ldo, n_shared,
                               AttnTrainArgs{});
        TC_CHECK_LAUNCH();
        return TC_OK;
    }
    dim3 grid((Lp_q / 128) * H * n_seq);
    if (dtype == TC_DTYPE_BF16)
        hipLaunchKernelGGL(attention_kernel<MmaBF16>, grid, dim3(256), 0, stream, (const char*)Q, (const char*)K,
                           (const char*)V, (char*)O, H, Lq, Lk, Lp_q, Lp_k, ldo, n_shared);
    else if (dtype == TC_DTYPE_BF16X3)      // fp32 images, products as split-bf16 triples (common.h MmaBF16x3)
        hipLaunchKernelGGL(attention_kernel<MmaBF16x3>, grid, dim3(256), 0, stream, (const char*)Q, (const char*)K,
                           (const char*)V, (char*)O, H, Lq, Lk, Lp_q, Lp_k, ldo, n_shared);
    else
        hipLaunchKernelGGL(attention_kernel<MmaF32>, grid, dim3(256), 0, stream, (const char*)Q, (const char*)K,
                           (const char*)V, (char*)O, H, Lq, Lk, Lp_q, Lp_k, ldo, n_shared);
    TC_CHECK_LAUNCH();
    return TC_OK;
}
