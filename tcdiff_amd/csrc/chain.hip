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
//
// Only the weights stream.  Structure:
//   * 8 waves, wave w owns output columns [64 w, 64 w + 64) of every 512-wide GEMM (= head w of Q / K / V) for all 64
//     rows; MFMA operand roles are swapped (A operand = weight rows, B operand = activation rows), so a lane holds 4
//     consecutive columns of ONE row per register quad: LayerNorm statistics are in-register sums + one cross-half
//     swap + an 8-wave exchange through LDS, and FiLM / residual / rotary / bf16 packing need no transposition.
//   * the weights of a chain are packed ON THE HOST (engine.py, once per checkpoint) into one linear stream per wave in
//     consumption order, in 2-KB stages that are already the LDS fragment image ([half][row][16 B]): a stage is two
//     1-KB global_load_lds pieces, each wave feeds a PRIVATE ring of 4 stages and consumes only what it loaded, so the
//     GEMM loops have NO workgroup barrier: one counted s_waitcnt vmcnt per stage.  The stream runs ahead across GEMM
//     and epilogue boundaries (the next GEMM's first stages land during the LayerNorm in front of it).
//   * activations live in LDS as [k-tile][64 rows][128 B] with the XOR chunk swizzle of common.h (tile_off).
// Barriers: one pair per LayerNorm (statistics exchange) and one per activation hand-off.
#include "common.h"
#include "tcdiff_hip.h"

#define CH_ABUF 0            // 64 KB  activation block [8 k-tiles][64][128 B]
#define CH_H1C 65536         // 32 KB  GELU(linear1) chunk [4 k-tiles][64][128 B]; epilogue scratch aliases it
#define CH_RING 98304        // 64 KB  8 private rings of 4 stages
#define CH_SMEM 163840
#define CH_NSLOT 4
#define CH_STAGE 2048

typedef const float* fptr;

struct WStream {
    const char* src;   // this wave's stream (wave-uniform)
    char* ring;        // this wave's ring
    unsigned issued;   // stages issued
    unsigned cons;     // stages consumed
    unsigned last;     // index of the last stage of the stream
};

// issue the next stage; past the end the last stage is re-read into a slot nobody reads again, which keeps the number
// of pieces in flight -- and with it the vmcnt arithmetic of ws_wait -- constant to the end of the stream
DEVINL void ws_issue(WStream& ws, int lane) {
    const unsigned st = ws.issued < ws.last ? ws.issued : ws.last;
    const unsigned off = st * CH_STAGE + (unsigned)lane * 16u;
    char* dst = ws.ring + (ws.issued % CH_NSLOT) * CH_STAGE;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slot's fragment reads have returned
    glds16(ws.src, off, dst);
    glds16(ws.src, off + 1024u, dst + 1024);
    ws.issued++;
}
// the oldest stage in flight has landed: all but the 2 x (CH_NSLOT - 1) youngest vector-memory operations are done
// (any other younger operation, e.g. an epilogue store, only makes the wait stricter)
DEVINL const char* ws_wait(const WStream& ws) {
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    return ws.ring + (ws.cons % CH_NSLOT) * CH_STAGE;
}
DEVINL void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

DEVINL f32x4_t ld4(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }
DEVINL void zero(f32x16_t& v) {
    const f32x16_t z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    v = z;
}

// acc[mi][ni] (rows 32 mi + r, columns 64 wave + 32 ni + ..) += act[64 x 16 nst] (k-steps kstep0.. of `abuf`) * W stage
DEVINL void phase_n512(f32x16_t (&acc)[2][2], const char* abuf, int kstep0, int nst, WStream& ws, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll 2
    for (int s = 0; s < nst; ++s) {
        const char* slot = ws_wait(ws);
        const int ks = kstep0 + s;
        const char* at = abuf + (ks >> 2) * 8192;
        const int ch = 2 * (ks & 3) + h;
        const u32x4 a0 = lds_frag(at, r, ch), a1 = lds_frag(at, 32 + r, ch);
        const u32x4 w0 = *reinterpret_cast<const u32x4*>(slot + h * 1024 + r * 16);
        const u32x4 w1 = *reinterpret_cast<const u32x4*>(slot + h * 1024 + (32 + r) * 16);
        MmaBF16::mma(acc[0][0], w0, a0);
        MmaBF16::mma(acc[0][1], w1, a0);
        MmaBF16::mma(acc[1][0], w0, a1);
        MmaBF16::mma(acc[1][1], w1, a1);
        ws_issue(ws, lane);
        ws.cons++;
    }
}
// linear1 chunk: acc[mi] (columns 32 wave + ..) += act[64 x 512] * W1 chunk; a stage = 2 k-steps of 32 weight rows
DEVINL void phase_ff1(f32x16_t (&acc)[2], const char* abuf, WStream& ws, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll 2
    for (int s = 0; s < 16; ++s) {
        const char* slot = ws_wait(ws);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int ks = 2 * s + k2;
            const char* at = abuf + (ks >> 2) * 8192;
            const int ch = 2 * (ks & 3) + h;
            const u32x4 a0 = lds_frag(at, r, ch), a1 = lds_frag(at, 32 + r, ch);
            const u32x4 w = *reinterpret_cast<const u32x4*>(slot + k2 * 1024 + h * 512 + r * 16);
            MmaBF16::mma(acc[0], w, a0);
            MmaBF16::mma(acc[1], w, a1);
        }
        ws_issue(ws, lane);
        ws.cons++;
    }
}

// LayerNorm statistics of the 64 rows over all 512 columns (two-pass, fp32): this lane's rows are 32 mi + r
DEVINL void row_stats(const f32x16_t (&acc)[2][2], float* scr, int wave, int lane, float eps, float (&mean)[2],
                      float (&rstd)[2]) {
    const int r = lane & 31;
    float s[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        float t = 0.0f;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int q = 0; q < 16; ++q) t += acc[mi][ni][q];
        s[mi] = t + other_half(t);
    }
    if (lane < 32) {
        scr[wave * 64 + r] = s[0];
        scr[wave * 64 + 32 + r] = s[1];
    }
    lds_barrier();
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += scr[w * 64 + 32 * mi + r];
        mean[mi] = t * (1.0f / 512.0f);
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        float t = 0.0f;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float d = acc[mi][ni][q] - mean[mi];
                t += d * d;
            }
        s[mi] = t + other_half(t);
    }
    if (lane < 32) {
        scr[512 + wave * 64 + r] = s[0];
        scr[512 + wave * 64 + 32 + r] = s[1];
    }
    lds_barrier();
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += scr[512 + w * 64 + 32 * mi + r];
        rstd[mi] = rsqrtf(t * (1.0f / 512.0f) + eps);
    }
}

// u = LayerNorm(acc) (optionally rotated) -> bf16 -> activation block in LDS (k = column)
template <bool ROT>
DEVINL void norm_to_lds(const f32x16_t (&acc)[2][2], const float (&mean)[2], const float (&rstd)[2], fptr g, fptr b,
                        fptr rope, int L, int m0, int M, char* abuf, int wave, int lane, uint2 (*keep)[2][4]) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        int m = m0 + 32 * mi + r;
        m = m < M ? m : M - 1;
        const int pos = m % L;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = 64 * wave + 32 * ni + 8 * gq + 4 * h;
                const f32x4_t g4 = ld4(g + n), b4 = ld4(b + n);
                float u[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) u[t] = (acc[mi][ni][4 * gq + t] - mean[mi]) * rstd[mi] * g4[t] + b4[t];
                if (keep) {   // the un-rotated image is needed later (V = norm1(x) W_v): keep it packed in registers
                    uint2 pk;
                    pk.x = pack_bf2(u[0], u[1]);
                    pk.y = pack_bf2(u[2], u[3]);
                    keep[mi][ni][gq] = pk;
                }
                if (ROT) {
                    const f32x4_t cs = ld4(rope + (long)pos * 512 + n);   // cos0 sin0 cos1 sin1
                    const float y0 = u[0] * cs[0] - u[1] * cs[1], y1 = u[1] * cs[0] + u[0] * cs[1];
                    const float y2 = u[2] * cs[2] - u[3] * cs[3], y3 = u[3] * cs[2] + u[2] * cs[3];
                    u[0] = y0; u[1] = y1; u[2] = y2; u[3] = y3;
                }
                uint2 pk;
                pk.x = pack_bf2(u[0], u[1]);
                pk.y = pack_bf2(u[2], u[3]);
                *reinterpret_cast<uint2*>(abuf + wave * 8192 + tile_off(32 * mi + r, 4 * ni + gq) + 8 * h) = pk;
            }
    }
}

// head-major scatter of a 512-wide projection (wave = head): model/model.py:78-80,92-95
DEVINL void store_heads(const f32x16_t (&acc)[2][2], void* base, float scale, int L, int Lp, int H, int m0, int M,
                        int wave, int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + 32 * mi + r;
        if (m >= M) continue;
        const int seq = m / L, tok = m % L;
        uint16_t* dst = reinterpret_cast<uint16_t*>(base) + (((long)seq * H + wave) * Lp + tok) * 64;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                uint2 pk;
                pk.x = pack_bf2(acc[mi][ni][4 * gq + 0] * scale, acc[mi][ni][4 * gq + 1] * scale);
                pk.y = pack_bf2(acc[mi][ni][4 * gq + 2] * scale, acc[mi][ni][4 * gq + 3] * scale);
                *reinterpret_cast<uint2*>(dst + 32 * ni + 8 * gq + 4 * h) = pk;
            }
    }
}

template <int MODE>
__global__ __launch_bounds__(512) void chain_kernel(tcdiff_chain_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * 64;
    const int M = a.M, L = a.L;
    char* abuf = smem + CH_ABUF;
    char* h1c = smem + CH_H1C;
    float* scr = reinterpret_cast<float*>(smem + CH_H1C);

    WStream ws;
    ws.src = reinterpret_cast<const char*>(a.wstream) + (long)wave * a.n_stages * CH_STAGE;
    ws.ring = smem + CH_RING + wave * (CH_NSLOT * CH_STAGE);
    ws.issued = 0;
    ws.cons = 0;
    ws.last = (unsigned)a.n_stages - 1;

    // ---- the block's input rows (attention output) -> LDS, then the first weight stages
#pragma unroll
    for (int kt = 0; kt < 8; ++kt)
        stage_glds<64, 8>(abuf + kt * 8192, reinterpret_cast<const char*>(a.A) + kt * TC_ROWB, 1024, m0, M, a.a_mod, wave,
                          lane);
#pragma unroll
    for (int i = 0; i < CH_NSLOT; ++i) ws_issue(ws, lane);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // the 8 activation pieces are older than the 8 weight pieces
    lds_barrier();

    f32x16_t acc[2][2];
    auto clear = [&]() {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) zero(acc[mi][ni]);
    };
    float mean[2], rstd[2];

    // ================= fc: LayerNorm(1e-6), FiLM, residual (model/model.py:103-106,171-173,327 / 334)
    clear();
    phase_n512(acc, abuf, 0, 32, ws, lane);
    lds_barrier();                     // every wave is out of the GEMM: scratch (and later the activation block) are free
    row_stats(acc, scr, wave, lane, a.ln_eps, mean, rstd);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + 32 * mi + r;
        const int mc = m < M ? m : M - 1;
        const int mr = a.xres_mod > 0 ? mc % a.xres_mod : mc;
        const float* fp = a.film + (long)(mc / L) * a.film_ld;
        const float* xr = a.xres + (long)mr * 512;
        float* xo = a.xout + (long)mc * 512;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = 64 * wave + 32 * ni + 8 * gq + 4 * h;
                const f32x4_t g4 = ld4(a.ln_g + n), b4 = ld4(a.ln_b + n), sc = ld4(fp + n), sh = ld4(fp + 512 + n);
                const f32x4_t x4 = ld4(xr + n);
                f32x4_t o;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float v = (acc[mi][ni][4 * gq + t] - mean[mi]) * rstd[mi] * g4[t] + b4[t];
                    v = (sc[t] + 1.0f) * v + sh[t];
                    v = x4[t] + v;
                    acc[mi][ni][4 * gq + t] = v;
                    o[t] = v;
                }
                if (m < M) *reinterpret_cast<f32x4_t*>(xo + n) = o;
            }
    }
    row_stats(acc, scr + 1024, wave, lane, a.n2_eps, mean, rstd);
    if (MODE == TC_CHAIN_A) {
        // norm2 + rotary (model/model.py:332,387) -> LDS -> Q = rot W_q^T / 8 (model/model.py:78,97)
        norm_to_lds<true>(acc, mean, rstd, a.n2_g, a.n2_b, a.rope, L, m0, M, abuf, wave, lane, nullptr);
        lds_barrier();
        clear();
        phase_n512(acc, abuf, 0, 32, ws, lane);
        store_heads(acc, a.q_out, a.scale_q, L, a.Lp, a.H, m0, M, wave, lane);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    // ================= feed-forward (model/model.py:338-339,399-401): norm3 -> LDS
    norm_to_lds<false>(acc, mean, rstd, a.n2_g, a.n2_b, nullptr, L, m0, M, abuf, wave, lane, nullptr);
    lds_barrier();
    clear();   // acc = linear2 accumulator
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
        f32x16_t a1[2];
        zero(a1[0]);
        zero(a1[1]);
        phase_ff1(a1, abuf, ws, lane);
        lds_barrier();                 // the previous chunk's linear2 reads of h1c (and the scratch reads) are done
        {
            const int nb = 256 * c + 32 * wave;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4_t b4 = ld4(a.b1 + nb + 8 * gq + 4 * h);
                    float v[4];
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = a1[mi][4 * gq + t] + b4[t];
                    act4_ct<ACT_GELU>(v, ACT_GELU);
                    uint2 pk;
                    pk.x = pack_bf2(v[0], v[1]);
                    pk.y = pack_bf2(v[2], v[3]);
                    // chunk column 32 wave + 8 gq + 4 h: k-tile wave / 2, 16-byte chunk 4 (wave & 1) + gq
                    *reinterpret_cast<uint2*>(h1c + (wave >> 1) * 8192 + tile_off(32 * mi + r, 4 * (wave & 1) + gq) +
                                              8 * h) = pk;
                }
        }
        lds_barrier();
        phase_n512(acc, h1c, 0, 16, ws, lane);
    }
    lds_barrier();                     // h1c is free: scratch
    // linear2 bias, FiLM, residual (the x this lane stored above), norm4 -> LDS
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + 32 * mi + r;
        const int mc = m < M ? m : M - 1;
        const float* fp = a.film3 + (long)(mc / L) * a.film_ld;
        const float* xr = a.xout + (long)mc * 512;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = 64 * wave + 32 * ni + 8 * gq + 4 * h;
                const f32x4_t b4 = ld4(a.b2 + n), sc = ld4(fp + n), sh = ld4(fp + 512 + n), x4 = ld4(xr + n);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float v = acc[mi][ni][4 * gq + t] + b4[t];
                    v = (sc[t] + 1.0f) * v + sh[t];
                    acc[mi][ni][4 * gq + t] = x4[t] + v;
                }
            }
    }
    row_stats(acc, scr, wave, lane, a.n4_eps, mean, rstd);
    norm_to_lds<false>(acc, mean, rstd, a.n4_g, a.n4_b, nullptr, L, m0, M, abuf, wave, lane, nullptr);
    lds_barrier();
    // ================= x' = linear3(norm4(x)) + b3, no residual (model/model.py:344)
    clear();
    phase_n512(acc, abuf, 0, 32, ws, lane);
    lds_barrier();
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
        const int m = m0 + 32 * mi + r;
        const int mc = m < M ? m : M - 1;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int n = 64 * wave + 32 * ni + 8 * gq + 4 * h;
                const f32x4_t b4 = ld4(a.b3 + n);
                f32x4_t o;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    acc[mi][ni][4 * gq + t] += b4[t];
                    o[t] = acc[mi][ni][4 * gq + t];
                }
                if (m < M) {
                    if (MODE == TC_CHAIN_B_LAST) {   // the final projection reads bf16 rows (model/model.py:623)
                        uint2 pk;
                        pk.x = pack_bf2(o[0], o[1]);
                        pk.y = pack_bf2(o[2], o[3]);
                        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(a.h_out) + (long)mc * 512 + n) = pk;
                    } else {
                        *reinterpret_cast<f32x4_t*>(a.xout + (long)mc * 512 + n) = o;
                    }
                }
            }
    }
    if (MODE == TC_CHAIN_B_LAST) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    // ================= next layer: norm1 + rotary -> Q, K ; norm1 -> V (model/model.py:326,374-383,78-80)
    row_stats(acc, scr, wave, lane, a.nn_eps, mean, rstd);
    uint2 keep[2][2][4];
    norm_to_lds<true>(acc, mean, rstd, a.nn_g, a.nn_b, a.rope, L, m0, M, abuf, wave, lane, keep);
    lds_barrier();
    clear();
    phase_n512(acc, abuf, 0, 32, ws, lane);
    store_heads(acc, a.q_out, a.scale_q, L, a.Lp, a.H, m0, M, wave, lane);
    clear();
    phase_n512(acc, abuf, 0, 32, ws, lane);
    store_heads(acc, a.k_out, 1.0f, L, a.Lp, a.H, m0, M, wave, lane);
    lds_barrier();                     // every wave is done with the rotated image
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                *reinterpret_cast<uint2*>(abuf + wave * 8192 + tile_off(32 * mi + r, 4 * ni + gq) + 8 * h) =
                    keep[mi][ni][gq];
    lds_barrier();
    clear();
    phase_n512(acc, abuf, 0, 32, ws, lane);
    store_heads(acc, a.v_out, 1.0f, L, a.Lp, a.H, m0, M, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int tcdiff_chain(const tcdiff_chain_args* a, hipStream_t stream) {
    if (!a || a->M <= 0 || a->L <= 0 || !a->A || !a->wstream) return TC_ERR_ARG;
    const int want = a->mode == TC_CHAIN_A ? 64 : (a->mode == TC_CHAIN_B ? 288 : (a->mode == TC_CHAIN_B_LAST ? 192 : -1));
    if (want < 0 || a->n_stages != want) return TC_ERR_ARG;
    if (!a->ln_g || !a->ln_b || !a->film || a->film_ld % 4 || !a->xres || !a->xout || !a->n2_g || !a->n2_b) return TC_ERR_ARG;
    if ((long)(a->a_mod > 0 ? a->a_mod : a->M) * 1024 >= (1L << 32)) return TC_ERR_ARG;
    const void* ptrs[] = {a->A, a->wstream, a->ln_g, a->ln_b, a->film, a->xres, a->xout, a->n2_g, a->n2_b, a->rope,
                          a->q_out, a->b1, a->b2, a->film3, a->n4_g, a->n4_b, a->b3, a->nn_g, a->nn_b, a->k_out,
                          a->v_out, a->h_out};
    for (const void* p : ptrs)
        if (p && !al16(p)) return TC_ERR_ALIGN;
    if (a->mode == TC_CHAIN_A) {
        if (!a->rope || !a->q_out || a->H != 8 || a->Lp <= 0) return TC_ERR_ARG;
    } else {
        if (!a->b1 || !a->b2 || !a->film3 || !a->n4_g || !a->n4_b || !a->b3) return TC_ERR_ARG;
        if (a->mode == TC_CHAIN_B &&
            (!a->rope || !a->nn_g || !a->nn_b || !a->q_out || !a->k_out || !a->v_out || a->H != 8 || a->Lp <= 0))
            return TC_ERR_ARG;
        if (a->mode == TC_CHAIN_B_LAST && !a->h_out) return TC_ERR_ARG;
    }
    static tc_dev_state dev_state;
    const int n_cu = tc_device_once(dev_state, [](int) {
        hipError_t e0 = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<TC_CHAIN_A>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, CH_SMEM);
        hipError_t e1 = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<TC_CHAIN_B>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, CH_SMEM);
        hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<TC_CHAIN_B_LAST>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, CH_SMEM);
        return e0 != hipSuccess ? e0 : (e1 != hipSuccess ? e1 : e2);
    });
    if (n_cu < 0) return n_cu;
    dim3 grid((a->M + 63) / 64);
    if (a->mode == TC_CHAIN_A)
        hipLaunchKernelGGL(chain_kernel<TC_CHAIN_A>, grid, dim3(512), CH_SMEM, stream, *a);
    else if (a->mode == TC_CHAIN_B)
        hipLaunchKernelGGL(chain_kernel<TC_CHAIN_B>, grid, dim3(512), CH_SMEM, stream, *a);
    else
        hipLaunchKernelGGL(chain_kernel<TC_CHAIN_B_LAST>, grid, dim3(512), CH_SMEM, stream, *a);
    TC_CHECK_LAUNCH();
    return TC_OK;
}
