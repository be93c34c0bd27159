// Row-block chain kernels (bf16): the row-local part of a decoder layer in ONE launch, gfx950.
//
// Everything between two attention calls of FiLMTransformerDecoderLayer (model/model.py:331-344) is row-local:
//   tail:  c = LayerNorm_1e-6(O_cross Wfc^T); x2 = x1 + (1+s2) c + h2                    (:103-106, :334)
//          f = W2 gelu(W1 LN3(x2) + b1) + b2;  x3 = x2 + (1+s3) f + h3                    (:338-339, :399-401)
//          x' = W3 LN4(x3) + b3   (no residual)                                           (:344)
//          h' = LN1'(x'), rot' = rotary(h')   (next layer's self-attention input)         (:326, :375)
//   head:  a = LayerNorm_1e-6(O_self Wfc^T); x1 = x0 + (1+s1) a + h1; q = rotary(LN2(x1)) Wq^T / 8   (:327, :332, :387)
// The unfused path (gemm_rowln / gemm_tile) writes and re-reads every intermediate through HBM/Infinity Cache and is
// bound by that traffic; here a workgroup owns 64 complete rows, keeps the fp32 residual in registers and the bf16
// GEMM operands in LDS, and only streams WEIGHTS: ~3 MB per row block, as one continuous sequence of 32 KB / 16 KB
// tiles through two LDS stages (global_load_lds_dwordx4, next tile always in flight, also across GEMM boundaries).
//
// LDS map (bytes), 160 KB:
//   [      0,  66560)  A    resident GEMM A operand, 64 rows x (1024 B + 16 B pad), bf16   (O tile -> LN3 out -> LN4 out)
//   [  66560,  83968)  H1   FFN hidden chunk, 64 rows x (256 B + 16 B pad), bf16
//   [  98304, 131072)  S1   weight stage 1 (32 KB)
//   [ 131072, 163840)  S0   weight stage 0 (32 KB)   -- never aliased
//   weight tiles: [256 rows][128 B] (32 KB) for the N = 512 GEMMs, [128 rows][128 B] (16 KB) for FFN1 chunks
//   [      0, 131072)  TILE fp32 64 x 512 accumulator dump (aliases A, H1, S1; only live between GEMMs)
// Every GEMM phase has an even number of weight tiles, so each phase starts in S0 and the first tile of the next
// phase can be prefetched into S0 while TILE is live.
#include "common.h"
#include "tcdiff_hip.h"

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

#define CH_A_OFF 0
#define CH_A_RS 1040
#define CH_H1_OFF 66560
#define CH_H1_RS 272
#define CH_S1_OFF 98304
#define CH_S0_OFF 131072
#define CH_SMEM 163840

#ifdef TC_STAMP
__device__ unsigned long long* g_ch_stamp = nullptr;
extern "C" int tcdiff_debug_chain_stamp_buffer(void* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_ch_stamp), &p, sizeof(p)) == hipSuccess ? 0 : -1;
}
#define CH_STAMP(i) do { if (threadIdx.x == 0 && g_ch_stamp) g_ch_stamp[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CH_STAMP(i) do { } while (0)
#endif

// ---- weight tile staging ------------------------------------------------------------------------------
// [ROWS][RB bytes] tile of a [N][K] bf16 weight, rows row0.., k-bytes kb0..: linear 1-KiB blocks in LDS, XOR swizzle on
// the source chunk.  RB = 128: 8 rows per block, chunk ^= (row>>1)&7 (common.h tile_off).  RB = 64: 16 rows per block,
// chunk ^= (row>>2)&3 (a ds_read_b128 lane group {0-3,12-15,20-27} then touches 16 distinct 16-B slots of the 256-B bank row).
template <int ROWS, int RB>
DEVINL void stage_w(char* stage, const char* W, long ldw_b, int row0, long kb0, int wave, int lane) {
#ifdef CH_ABLATE_DMA
    return;
#endif
    constexpr int RPB = 1024 / RB;               // rows per 1-KiB block
    constexpr int NBLK = ROWS / RPB;
    constexpr int PER = NBLK / 8;
    static_assert(NBLK % 8 == 0, "blocks must divide over 8 waves");
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int blk = wave * PER + i;
        const int row = blk * RPB + (RB == 128 ? (lane >> 3) : (lane >> 2));
        const int slot = RB == 128 ? (lane & 7) : (lane & 3);
        const int chunk = RB == 128 ? (slot ^ ((row >> 1) & 7)) : (slot ^ ((row >> 2) & 3));
        const char* g = W + (long)(row0 + row) * ldw_b + kb0 + chunk * 16;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)(stage + blk * 1024), 16, 0, 0);
    }
}
template <int RB>
DEVINL u32x4 frag_w(const char* stage, int row, int chunk) {
    const int sw = RB == 128 ? ((row >> 1) & 7) : ((row >> 2) & 3);
    return *reinterpret_cast<const u32x4*>(stage + row * RB + ((chunk ^ sw) << 4));
}
// resident A operand: padded row-major, 16-B chunk c of row r at r*RS + c*16 (RS = 16 mod 256: 16 consecutive rows hit
// 16 distinct slots)
DEVINL u32x4 frag_a(const char* base, int rs, int row, int chunk) {
    return *reinterpret_cast<const u32x4*>(base + row * rs + chunk * 16);
}

#ifdef CH_ABLATE_MMA
#define CH_MMA(acc, a, b) asm volatile("" :: "v"(a), "v"(b))
#else
#define CH_MMA(acc, a, b) MmaBF16::mma(acc, a, b)
#endif

struct RowRegs {
    f32x4_t a[8], b[8];   // this wave's 8 rows: columns [4l,4l+4) and [256+4l,256+4l+4)
};

DEVINL void ln_row(f32x4_t& va, f32x4_t& vb, float eps, const f32x4_t& ga, const f32x4_t& gb, const f32x4_t& ba,
                   const f32x4_t& bb) {
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
}
DEVINL f32x4_t ld4(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }
DEVINL void st_bf4(char* p, const f32x4_t& v) {
    uint2 pk;
    pk.x = pack_bf2(v[0], v[1]);
    pk.y = pack_bf2(v[2], v[3]);
    *reinterpret_cast<uint2*>(p) = pk;
}
DEVINL f32x4_t rot4(const f32x4_t& v, const f32x4_t& cs) {  // cs = cos0 sin0 cos1 sin1
    f32x4_t y;
    y[0] = v[0] * cs[0] - v[1] * cs[1]; y[1] = v[1] * cs[0] + v[0] * cs[1];
    y[2] = v[2] * cs[2] - v[3] * cs[3]; y[3] = v[3] * cs[2] + v[2] * cs[3];
    return y;
}

// One staged [256][128 B] weight tile against the resident A operand (4 k-steps of 2 MFMAs).
DEVINL void half_tile(f32x16_t& acc0, f32x16_t& acc1, const char* stage, const char* abase, int a_rs, int achunk, int wm,
                      int wn, int r, int h) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const u32x4 fa = frag_a(abase, a_rs, wm * 32 + r, achunk + 2 * ks + h);
        CH_MMA(acc0, fa, frag_w<128>(stage, wn * 64 + r, 2 * ks + h));
        CH_MMA(acc1, fa, frag_w<128>(stage, wn * 64 + 32 + r, 2 * ks + h));
    }
}

// One GEMM phase of the chain: acc[4] (64 rows x 512 cols) += A(resident, k-bytes akb0..) * W[512 rows][K]^T.
// W streams as tiles of [256 rows][128 B] (full 128-B lines: a 64-B-row tile uses half of every line it touches and
// halves the effective L2 rate): k-tile kt, column half hf -> tile index 2*kt + hf, so half 0 always sits in S0 and
// half 1 in S1.  Wave (wm, wn) owns rows wm*32.. and, in each half, columns hf*256 + wn*64 + {0..63}:
// acc[2*hf + j] <-> column hf*256 + wn*64 + j*32 + (lane & 31).
// Protocol: on entry tile 0 of this phase is already in flight into S0; `issue_next` is called once, during the last
// tile, to start the first tile of the following phase (into S0).
template <int NKT, class NextFn>
DEVINL void gemm_n512(f32x16_t (&acc)[4], char* smem, const char* abase, int a_rs, int achunk0, const char* W, long ldw_b,
                      long wkb0, int wave, int lane, NextFn issue_next) {
    const int wm = wave >> 2, wn = wave & 3, r = lane & 31, h = lane >> 5;
    char* S0 = smem + CH_S0_OFF;
    char* S1 = smem + CH_S1_OFF;
#pragma unroll 1
    for (int kt = 0; kt < NKT; ++kt) {
        // ---- half 0 (S0); stream half 1 of this k-tile into S1 meanwhile
        sync_dma();
        stage_w<256, 128>(S1, W, ldw_b, 256, wkb0 + (long)kt * 128, wave, lane);
        half_tile(acc[0], acc[1], S0, abase, a_rs, achunk0 + kt * 8, wm, wn, r, h);
        // ---- half 1 (S1); stream half 0 of the next k-tile (or the next phase's first tile) into S0
        sync_dma();
        if (kt + 1 < NKT) stage_w<256, 128>(S0, W, ldw_b, 0, wkb0 + (long)(kt + 1) * 128, wave, lane);
        else issue_next();
        half_tile(acc[2], acc[3], S1, abase, a_rs, achunk0 + kt * 8, wm, wn, r, h);
    }
}

DEVINL void dump_tile(const f32x16_t (&acc)[4], float* tile, int wave, int lane) {
    const int wm = wave >> 2, wn = wave & 3, r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int n = (a >> 1) * 256 + wn * 64 + (a & 1) * 32 + r;
#pragma unroll
        for (int q = 0; q < 16; ++q) tile[(wm * 32 + acc_row(q, h)) * 512 + n] = acc[a][q];
    }
}
DEVINL void zero_acc(f32x16_t (&acc)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.0f;
}

// =================================================================================================
// layer tail
// =================================================================================================
__global__ __launch_bounds__(512) void chain_tail_kernel(tcdiff_tail_args e) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * 64;
    const int M = e.M;
    const int c0 = 4 * lane, c1 = 256 + 4 * lane;
    float* tile = reinterpret_cast<float*>(smem);
    char* Abuf = smem + CH_A_OFF;
    char* H1 = smem + CH_H1_OFF;
    const char* Wfc = reinterpret_cast<const char*>(e.Wfc);
    const char* W1 = reinterpret_cast<const char*>(e.W1);
    const char* W2 = reinterpret_cast<const char*>(e.W2);
    const char* W3 = reinterpret_cast<const char*>(e.W3);

    // ---- prologue: O tile (64 x 512 bf16) -> A (one DMA instruction per row), first Wfc tile -> S0, residual rows -> regs
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wave * 8 + i;
        int m = m0 + row;
        m = m < M ? m : M - 1;
        const char* g = reinterpret_cast<const char*>(e.O) + (long)m * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)g, (lds_void_t*)(Abuf + row * CH_A_RS), 16, 0, 0);
    }
    stage_w<256, 128>(smem + CH_S0_OFF, Wfc, 1024, 0, 0, wave, lane);
    RowRegs x;   // residual stream of this wave's 8 rows, fp32, lives in registers for the whole kernel
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        int m = m0 + wave * 8 + rr;
        m = m < M ? m : M - 1;
        x.a[rr] = ld4(e.xres + (long)m * 512 + c0);
        x.b[rr] = ld4(e.xres + (long)m * 512 + c1);
    }

    f32x16_t acc[4];
    CH_STAMP(0);
#ifdef TC_STAMP
    if (threadIdx.x == 0 && g_ch_stamp) g_ch_stamp[blockIdx.x * 16 + 8] = __builtin_amdgcn_s_memtime();
#endif
    // ---- GEMM fc: O Wfc^T ---------------------------------------------------------------------------
    zero_acc(acc);
    gemm_n512<8>(acc, smem, Abuf, CH_A_RS, 0, Wfc, 1024, 0, wave, lane,
                 [&] { stage_w<128, 128>(smem + CH_S0_OFF, W1, 1024, 0, 0, wave, lane); });
    sync_dma();                 // all waves done reading A and the stages (TILE aliases them); DMA + prefetches retired
    CH_STAMP(1);
    dump_tile(acc, tile, wave, lane);
    __syncthreads();
    {   // row phase 1: post-LN (1e-6), FiLM2, residual -> x2 (registers); LN3 -> A
        RowRegs v;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            v.a[rr] = ld4(tile + (wave * 8 + rr) * 512 + c0);
            v.b[rr] = ld4(tile + (wave * 8 + rr) * 512 + c1);
        }
        __syncthreads();        // TILE fully consumed: A may be overwritten
        const f32x4_t ga = ld4(e.lnp_g + c0), gb = ld4(e.lnp_g + c1), ba = ld4(e.lnp_b + c0), bb = ld4(e.lnp_b + c1);
        const f32x4_t g3a = ld4(e.ln3_g + c0), g3b = ld4(e.ln3_g + c1), b3a = ld4(e.ln3_b + c0), b3b = ld4(e.ln3_b + c1);
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            int m = m0 + wave * 8 + rr;
            m = m < M ? m : M - 1;
            const float* fp = e.film2 + (long)(m / e.L) * e.film_ld;
            ln_row(v.a[rr], v.b[rr], 1e-6f, ga, gb, ba, bb);
            x.a[rr] = x.a[rr] + ((ld4(fp + c0) + 1.0f) * v.a[rr] + ld4(fp + 512 + c0));
            x.b[rr] = x.b[rr] + ((ld4(fp + c1) + 1.0f) * v.b[rr] + ld4(fp + 512 + c1));
            f32x4_t ua = x.a[rr], ub = x.b[rr];
            ln_row(ua, ub, 1e-5f, g3a, g3b, b3a, b3b);
            st_bf4(Abuf + (wave * 8 + rr) * CH_A_RS + c0 * 2, ua);
            st_bf4(Abuf + (wave * 8 + rr) * CH_A_RS + c1 * 2, ub);
        }
    }

    CH_STAMP(2);
    // ---- FFN: 8 hidden chunks of 128: h1c = gelu(A W1c^T + b1c) -> H1;  acc += h1c W2[:, chunk]^T ---------------
    zero_acc(acc);
    {
        const int wm = wave >> 2, wn = wave & 3, r = lane & 31, h = lane >> 5;
#pragma unroll 1
        for (int c = 0; c < 8; ++c) {
            // GEMM1c: 64 x 128 output, wave = 32 rows x 32 cols; roles swapped: lane = row, registers = 4 consecutive cols
            f32x16_t a1;
#pragma unroll
            for (int q = 0; q < 16; ++q) a1[q] = 0.0f;
#pragma unroll 1
            for (int t = 0; t < 8; ++t) {
                sync_dma();
                char* cur = smem + ((t & 1) ? CH_S1_OFF : CH_S0_OFF);
                char* nxt = smem + ((t & 1) ? CH_S0_OFF : CH_S1_OFF);
                if (t + 1 < 8) stage_w<128, 128>(nxt, W1, 1024, c * 128, (long)(t + 1) * 128, wave, lane);
                else stage_w<256, 128>(nxt, W2, 2048, 0, (long)c * 256, wave, lane);  // first W2 tile of this chunk
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const u32x4 fa = frag_a(Abuf, CH_A_RS, wm * 32 + r, t * 8 + 2 * ks + h);
                    const u32x4 fw = frag_w<128>(cur, wn * 32 + r, 2 * ks + h);
                    CH_MMA(a1, fw, fa);
                }
            }
            // epilogue: + b1, GELU, bf16 -> H1[row = wm*32 + r][col = wn*32 + 8g + 4h + t]
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wn * 32 + 8 * g + 4 * h;
                const f32x4_t bq = ld4(e.b1 + c * 128 + nl);
                f32x4_t v;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = gelu_erf(a1[4 * g + t] + bq[t]);
                st_bf4(H1 + (wm * 32 + r) * CH_H1_RS + nl * 2, v);
            }
            // GEMM2c: K = 128 of this chunk = 2 k-tiles x 2 column halves of [256][128 B]
            gemm_n512<2>(acc, smem, H1, CH_H1_RS, 0, W2, 2048, (long)c * 256, wave, lane, [&] {
                if (c + 1 < 8) stage_w<128, 128>(smem + CH_S0_OFF, W1, 1024, (c + 1) * 128, 0, wave, lane);
                else stage_w<256, 128>(smem + CH_S0_OFF, W3, 1024, 0, 0, wave, lane);   // first linear3 tile
            });
        }
    }
    sync_dma();        // retire every DMA / prefetch (their dummy landing area is aliased by TILE)
    CH_STAMP(3);
    dump_tile(acc, tile, wave, lane);
    __syncthreads();
    {   // row phase 2: + b2, FiLM3, residual -> x3; LN4 -> A
        RowRegs v;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            v.a[rr] = ld4(tile + (wave * 8 + rr) * 512 + c0);
            v.b[rr] = ld4(tile + (wave * 8 + rr) * 512 + c1);
        }
        __syncthreads();
        const f32x4_t b2a = ld4(e.b2 + c0), b2b = ld4(e.b2 + c1);
        const f32x4_t g4a = ld4(e.ln4_g + c0), g4b = ld4(e.ln4_g + c1), b4a = ld4(e.ln4_b + c0), b4b = ld4(e.ln4_b + c1);
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            int m = m0 + wave * 8 + rr;
            m = m < M ? m : M - 1;
            const float* fp = e.film3 + (long)(m / e.L) * e.film_ld;
            x.a[rr] = x.a[rr] + ((ld4(fp + c0) + 1.0f) * (v.a[rr] + b2a) + ld4(fp + 512 + c0));
            x.b[rr] = x.b[rr] + ((ld4(fp + c1) + 1.0f) * (v.b[rr] + b2b) + ld4(fp + 512 + c1));
            f32x4_t ua = x.a[rr], ub = x.b[rr];
            ln_row(ua, ub, 1e-5f, g4a, g4b, b4a, b4b);
            st_bf4(Abuf + (wave * 8 + rr) * CH_A_RS + c0 * 2, ua);
            st_bf4(Abuf + (wave * 8 + rr) * CH_A_RS + c1 * 2, ub);
        }
    }
    CH_STAMP(4);
    // ---- linear3 ---------------------------------------------------------------------------------------
    zero_acc(acc);
    gemm_n512<8>(acc, smem, Abuf, CH_A_RS, 0, W3, 1024, 0, wave, lane, [] {});
    sync_dma();
    CH_STAMP(5);
    dump_tile(acc, tile, wave, lane);
    __syncthreads();
    {   // row phase 3: + b3 -> x' (stored); next layer's LN1 + rotary -> h', rot'
        const f32x4_t b3a = ld4(e.b3 + c0), b3b = ld4(e.b3 + c1);
        f32x4_t gna, gnb, bna, bnb;
        if (e.ln1n_g) {
            gna = ld4(e.ln1n_g + c0); gnb = ld4(e.ln1n_g + c1); bna = ld4(e.ln1n_b + c0); bnb = ld4(e.ln1n_b + c1);
        }
#pragma unroll 2
        for (int rr = 0; rr < 8; ++rr) {
            const int m = m0 + wave * 8 + rr;
            if (m >= M) break;
            f32x4_t va = ld4(tile + (wave * 8 + rr) * 512 + c0) + b3a;
            f32x4_t vb = ld4(tile + (wave * 8 + rr) * 512 + c1) + b3b;
            if (e.xout) {
                *reinterpret_cast<f32x4_t*>(e.xout + (long)m * 512 + c0) = va;
                *reinterpret_cast<f32x4_t*>(e.xout + (long)m * 512 + c1) = vb;
            }
            char* hrow = reinterpret_cast<char*>(e.hout) + (long)m * 1024;
            if (!e.ln1n_g) {   // last layer: bf16 copy of x' (A operand of final_layer)
                st_bf4(hrow + c0 * 2, va);
                st_bf4(hrow + c1 * 2, vb);
                continue;
            }
            ln_row(va, vb, 1e-5f, gna, gnb, bna, bnb);
            st_bf4(hrow + c0 * 2, va);
            st_bf4(hrow + c1 * 2, vb);
            const int pos = m % e.L;
            char* rrow = reinterpret_cast<char*>(e.rout) + (long)m * 1024;
            st_bf4(rrow + c0 * 2, rot4(va, ld4(e.rope + (long)pos * 512 + c0)));
            st_bf4(rrow + c1 * 2, rot4(vb, ld4(e.rope + (long)pos * 512 + c1)));
        }
    }
    CH_STAMP(6);
#ifdef TC_STAMP
    if (threadIdx.x == 0 && g_ch_stamp) g_ch_stamp[blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memtime();
#endif
}

static bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

extern "C" int tcdiff_chain_tail(const tcdiff_tail_args* a, hipStream_t stream) {
    if (!a || a->M <= 0 || a->L <= 0) return TC_ERR_ARG;
    const void* req[] = {a->O, a->Wfc, a->lnp_g, a->lnp_b, a->film2, a->film3, a->xres, a->ln3_g, a->ln3_b, a->W1, a->b1,
                         a->W2, a->b2, a->ln4_g, a->ln4_b, a->W3, a->b3, a->hout};
    for (const void* p : req) {
        if (!p) return TC_ERR_ARG;
        if (!al16(p)) return TC_ERR_ALIGN;
    }
    if (a->ln1n_g && (!a->ln1n_b || !a->rout || !a->rope)) return TC_ERR_ARG;
    if (a->film_ld % 4) return TC_ERR_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(chain_tail_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, CH_SMEM);
        attr_set = true;
    }
    hipLaunchKernelGGL(chain_tail_kernel, dim3((a->M + 63) / 64), dim3(512), CH_SMEM, stream, *a);
    TC_CHECK_LAUNCH();
    return TC_OK;
}
