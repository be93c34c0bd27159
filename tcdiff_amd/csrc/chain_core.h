// Building blocks of the row-block kernels (chain.hip: the decoder-layer chains of the sampler; gemm_rows.hip: the training
// step's GEMMs): the per-wave weight stream, the 16x16x32 MFMA phase over an LDS-resident activation block, lane-group
// exchanges, LayerNorm statistics, the column-blocked row pipeline and the head-major scatter.  See chain.hip's header
// comment for the structure they implement.
#pragma once
#include "common.h"
#include "tcdiff_hip.h"

// LDS map.  The small constant areas come first so that their reads are `base VGPR + 16-bit immediate`.
#define CH_FILM 0            //  8 KB  FiLM (scale + 1 | shift) rows of the <= 2 sequences this block touches, for the next epilogue
#define CH_VEC 8192          // 12 KB  six 512-float vectors (LayerNorm weights, biases) of the next epilogue(s)
#define CH_SCR 20480         //  8 KB  LayerNorm statistics exchange: 2 x [8 waves][64 rows] float2
#define CH_ABUF 28672        // 64 KB  activation block [8 k-tiles][64][128 B]
#define CH_H1C 94208         // 32 KB  GELU(linear1) chunk [4 k-tiles][64][128 B]
#define CH_ABUF2 94208       // 64 KB  second activation block (un-rotated norm1 image for V); overlays the dead h1 chunk
#define CH_STG7 159744       //  4 KB  eighth staging slot of store_heads (slots 0-6: the first 28 KB)
#define CH_SMEM 163840
#define CH_D 4               // weight stages in flight per wave (registers): 4 x 4 KB x 8 waves = 128 KB per CU
#define CH_STAGE 4096
#ifndef CH_D8
#define CH_D8 4              // ... of the 4-wave form (NT = 8: 8-KB stages, 32 registers each): 4 x 8 KB x 4 waves = 128 KB per CU
#endif
// NT = 16-column n-tiles per wave: 4 -> 8 waves x 64 columns (two waves per SIMD, <= 256 registers each), 8 -> 4 waves x 128
// columns (ONE wave per SIMD with the 512-register budget: every LDS activation fragment feeds 8 MFMAs instead of 4, the
// LayerNorm exchange is between 4 waves, and a wave's own MFMA stream is long enough to carry independent VALU work).
template <int NT> struct ChW {
    static constexpr int NW = 32 / NT;                 // waves per workgroup
    static constexpr int D = NT == 8 ? CH_D8 : CH_D;   // ring depth in stages
    static constexpr int STAGE = 1024 * NT;            // bytes per stage = one 32-deep k-step of the wave's 16 NT weight rows
};

typedef const float* fptr;
// accumulators: f32x4_t acc[4 n-tiles][MT m-tiles of 16 rows]; MT = 4 (64-row blocks: the benchmark), 2 or 1 (small jobs:
// more, smaller row blocks so that every CU gets one -- a block costs one pass over the layer's weights whatever its rows)

DEVINL void mma16(f32x4_t& acc, const u32x4& a, const u32x4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}

// A wave's weight stream.  The fragments of a stage are private to the wave (it owns the output columns they produce),
// so they never need LDS: a stage is four coalesced 1-KB global loads straight into the registers the MFMAs read, and
// the ring of CH_D stages in flight is a register array indexed at compile time (every loop over it is unrolled).
// The compiler counts vmcnt for these loads itself.  Past the end of the stream the last stage is re-read (never used).
template <int NT = 4>
struct WStreamT {
    __amdgpu_buffer_rsrc_t rsrc;   // this wave's stream as a raw buffer: a stage address is SGPR descriptor + SGPR stage
    unsigned voff;                 // offset + this one VGPR (lane * 16)
    unsigned pos;      // stages consumed so far (wave-uniform)
    unsigned last;     // index of the last stage
    u32x4 w[ChW<NT>::D][NT];
};
typedef WStreamT<4> WStream;
template <int NT>
DEVINL void ws_load(WStreamT<NT>& ws, int slot, unsigned stage) {
    const unsigned st = stage < ws.last ? stage : ws.last;
    const unsigned so = st * (unsigned)ChW<NT>::STAGE;       // scalar
#pragma unroll
    for (int i = 0; i < NT; ++i)
        ws.w[slot][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ws.rsrc, ws.voff + 1024u * i, so, 0));
}
// IR-level fence for memory operations + machine-scheduler fence for everything: keeps an unrolled epilogue loop one
// iteration at a time (see the fc epilogue)
#define CH_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
DEVINL void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Fresh copies of the lane / wave index that the compiler cannot relate to earlier ones: every phase derives its LDS
// and global addresses from its own copy, so address arithmetic is recomputed per phase (a few VALU ops) instead of
// being hoisted to the top of the kernel and kept alive across it -- which, with 64 accumulator + 64 ring registers
// resident, spills, and every scratch reload in an epilogue is a full memory round trip.
DEVINL int fresh_v(int x) {
    asm volatile("" : "+v"(x));
    return x;
}
DEVINL int fresh_s(int x) {
    asm volatile("" : "+s"(x));
    return x;
}

DEVINL f32x4_t ld4(const float* p) { return *reinterpret_cast<const f32x4_t*>(p); }
template <int MT, int NT>
DEVINL void zero(f32x4_t (&a)[NT][MT]) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) a[nt][mt] = f32x4_t{0, 0, 0, 0};
}

// ---- lane-group exchanges (the four 16-lane groups g of a wave hold different columns of the same rows) -----------------
// v_permlane16_swap vdst, src: lanes 16-31 / 48-63 of vdst swap with lanes 0-15 / 32-47 of src;
// v_permlane32_swap vdst, src: lanes 32-63 of vdst swap with lanes 0-31 of src (cdna_hip_programming.md T21).
DEVINL void swap16(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    a = __builtin_bit_cast(float, (unsigned)r[0]);
    b = __builtin_bit_cast(float, (unsigned)r[1]);
}
DEVINL void swap32(float& a, float& b) {
    auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
    a = __builtin_bit_cast(float, (unsigned)r[0]);
    b = __builtin_bit_cast(float, (unsigned)r[1]);
}
// reduce-scatter: a[t] = this lane's partial of row tile t (over its lane group's columns) -> the sum over the four lane
// groups of row tile g, in lane group g (3 swaps + 3 adds for 4 values)
DEVINL float rs4_sum(float a0, float a1, float a2, float a3) {
    swap16(a0, a1);          // a0 = [a0.g0, a1.g0, a0.g2, a1.g2], a1 = [a0.g1, a1.g1, a0.g3, a1.g3]
    swap16(a2, a3);
    float p01 = a0 + a1, p23 = a2 + a3;      // p01 = [a0(g0+g1), a1(g0+g1), a0(g2+g3), a1(g2+g3)]
    swap32(p01, p23);        // p01 = [a0(g0+g1), a1(g0+g1), a2(g0+g1), a3(g0+g1)], p23 = the (g2+g3) halves
    return p01 + p23;
}
// all-gather: v = the value of row tile g in lane group g -> out[t] = row tile t's value, in every lane group
DEVINL void ag4(float v, float (&out)[4]) {
    float r0 = v, r1 = v;
    swap16(r0, r1);          // r0 = [v.g0, v.g0, v.g2, v.g2], r1 = [v.g1, v.g1, v.g3, v.g3]
    float q0 = r0, q2 = r0, q1 = r1, q3 = r1;
    swap32(q0, q2);          // q0 = v.g0 everywhere, q2 = v.g2 everywhere
    swap32(q1, q3);
    out[0] = q0; out[1] = q1; out[2] = q2; out[3] = q3;
}
// maximum over the four lane groups, in every lane
DEVINL float ar4_max(float v) {
    float a = v, b = v;
    swap16(a, b);
    v = fmaxf(a, b);
    a = v; b = v;
    swap32(a, b);
    return fmaxf(a, b);
}
DEVINL float ar4_sum(float v) {
    float a = v, b = v;
    swap16(a, b);
    v = a + b;
    a = v; b = v;
    swap32(a, b);
    return a + b;
}

// every accumulator tile through an (empty) asm statement: orders the MFMAs in front of it before everything behind it
// (NT = 8: the 128 accumulator registers of a 4-wave block live in the ACCUMULATOR half of the wave's 512 registers -- the
// fence must name them with the "a" constraint, or every stage copies all of them to VGPRs and back)
#define CH_ACC_FENCE_BODY(C_)                                                                                                    \
    if constexpr (MT == 4) {                                                                                                     \
        _Pragma("unroll") for (int nt = 0; nt < NT; nt += 2)                                                                     \
            asm volatile("" : C_(a[nt][0]), C_(a[nt][1]), C_(a[nt][2]), C_(a[nt][3]), C_(a[nt + 1][0]), C_(a[nt + 1][1]),       \
                         C_(a[nt + 1][2]), C_(a[nt + 1][3]));                                                                    \
    } else if constexpr (MT == 2) {                                                                                              \
        _Pragma("unroll") for (int nt = 0; nt < NT; nt += 4)                                                                     \
            asm volatile("" : C_(a[nt][0]), C_(a[nt][1]), C_(a[nt + 1][0]), C_(a[nt + 1][1]), C_(a[nt + 2][0]),                  \
                         C_(a[nt + 2][1]), C_(a[nt + 3][0]), C_(a[nt + 3][1]));                                                  \
    } else {                                                                                                                     \
        _Pragma("unroll") for (int nt = 0; nt < NT; nt += 4)                                                                     \
            asm volatile("" : C_(a[nt][0]), C_(a[nt + 1][0]), C_(a[nt + 2][0]), C_(a[nt + 3][0]));                               \
    }
#define CH_CV(x) "+v"(x)
#define CH_CA(x) "+a"(x)
template <int MT, int NT>
DEVINL void acc_fence(f32x4_t (&a)[NT][MT]) {
    if constexpr (NT == 8 && MT > 1) {
        CH_ACC_FENCE_BODY(CH_CA)
    } else {
        CH_ACC_FENCE_BODY(CH_CV)
    }
}
// the next stage's B fragments through the stage fence (see phase_n512)
template <int MT>
DEVINL void frag_fence(u32x4 (&n)[MT]) {
    if constexpr (MT == 4) asm volatile("" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]) : : "memory");
    else if constexpr (MT == 2) asm volatile("" : "+v"(n[0]), "+v"(n[1]) : : "memory");
    else asm volatile("" : "+v"(n[0]) : : "memory");
}

// LDS addresses of this lane's B fragments: row 16 mt + c of a [64][128 B] k-tile, chunk 4 (ks & 1) + PI(g) of k-step ks.
// The two bases (even / odd k-step) INCLUDE the activation block's LDS address and are opaque to the compiler, so that every
// fragment read is `base VGPR + 16-bit immediate` ((ks >> 1) * 8192 + mt * 2048 <= 63488): folded into the immediate, the
// block's own offset (28 KB ..) pushes the later k-tiles past 65535 and every one of them costs an address VGPR.
typedef const __attribute__((address_space(3))) u32x4 lds_u32x4;
struct FragOff { unsigned e, o; };
DEVINL FragOff frag_off(const char* abuf, int lane) {
    const int c = lane & 15, g = lane >> 4;
    const int pg = (0x9C >> (2 * g)) & 3;                   // PI = (0, 3, 1, 2)
    const int sw = tile_swz(c);                             // (row >> 1) & 7 of row 16 mt + c does not depend on mt
    const unsigned base = (unsigned)reinterpret_cast<uintptr_t>(abuf);      // the low half of a generic LDS pointer is the LDS address
    FragOff f;
    f.e = base + c * TC_ROWB + ((pg ^ sw) << 4);
    f.o = base + c * TC_ROWB + (((4 + pg) ^ sw) << 4);
    asm volatile("" : "+v"(f.e), "+v"(f.o));
    return f;
}
DEVINL u32x4 frag_rd(const FragOff& f, int ks, int mt) {
    return *reinterpret_cast<lds_u32x4*>((uintptr_t)(((ks & 1) ? f.o : f.e) + (unsigned)((ks >> 1) * 8192 + mt * 2048)));
}

// acc[nt][mt] (rows 16 mt + c, columns 64 wave + 16 nt + ..) += act[64 x 32 NST] (k-steps 0.. of `abuf`) * W stages;
// a stage = one 32-deep k-step of the wave's 64 weight rows: fragment nt = weight rows 16 nt + c.  NST % CH_D == 0.
// Fully unrolled (NST <= 32 stage bodies): a rolled loop carries the ring through a phi, and hipcc placed a register
// copy of the most recently loaded slot at the loop header -- i.e. `s_waitcnt vmcnt(0)`, a full drain of the wave's
// weight stream, every CH_D stages.  The activation fragments of stage ks + 1 are read before the MFMAs of stage ks.
// TAIL: the launch's last phase -- its last CH_D stages refill nothing (there is nothing behind them).
// SWAP: the MFMA operands change places -- acc[nt][mt] then holds the TRANSPOSED tiles (lane (c, g): rows 16 mt + 4 g + j, column
// 64 wave + 16 nt + c), which is the order a V^T fragment of the attention wants its keys in (chain.hip, store_vfrag).
template <int NST, bool TAIL = false, int MT = 4, int NT = 4, bool SWAP = false>
DEVINL void phase_n512(f32x4_t (&acc)[NT][MT], const char* abuf, WStreamT<NT>& ws, int lane,
                       unsigned long long* stage_stamps = nullptr) {     // (diagnostic builds: s_memtime after every stage)
    constexpr int RD = ChW<NT>::D;
    static_assert(NST % RD == 0, "a phase starts and ends at ring slot 0");
    lane = fresh_v(lane);
    const FragOff fo = frag_off(abuf, lane);
    u32x4 b[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) b[mt] = frag_rd(fo, 0, mt);
    const unsigned base = ws.pos;
    FragOff fo2 = fo;                  // K = 1024 (TC_CHAIN_FRONT): k-tiles 8 .. 15 from a second pair of bases
    if (NST > 16) {
        fo2.e += 65536u;
        fo2.o += 65536u;
        asm volatile("" : "+v"(fo2.e), "+v"(fo2.o));
    }
#pragma unroll
    for (int ks = 0; ks < NST; ++ks) {
        const int i = ks % RD;
        u32x4 wv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wv[nt] = ws.w[i][nt];
        if (!(TAIL && ks + RD >= NST)) ws_load(ws, i, base + ks + RD);
        u32x4 nb[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) nb[mt] = b[mt];
        if (ks + 1 < NST) {
            const FragOff& fn = ks + 1 < 16 ? fo : fo2;
            const int kl = (ks + 1) & 15;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) nb[mt] = frag_rd(fn, kl, mt);
        }
        // Stage order: the next stage's LDS reads and the refill are ISSUED, then this stage's MFMAs run (the ~100
        // cycles of LDS latency pass under them even when the wave is alone on its SIMD), then the fence.  Left to
        // itself hipcc schedules read, wait, use.
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (SWAP) mma16(acc[nt][mt], b[mt], wv[nt]);
                else mma16(acc[nt][mt], wv[nt], b[mt]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // ... and the MFMAs do not sink below the next stage's reads either
        // Stage fence.  Memory clobber: the refill stays in its own stage, the stream never drains.  The next stage's
        // fragments pass THROUGH it, so the next stage's MFMAs cannot be pulled up to right behind their reads; the
        // accumulators pass through it too: an MFMA has no side effect, and instruction selection otherwise defers whole
        // stages of them past the following stages' loads (seen in the listing: empty stages, then 40 MFMAs in a row with
        // three stages of fragments and refills live -- 232 VGPRs and four ring slots spilled behind `s_waitcnt vmcnt(0)`).
        acc_fence(acc);
        frag_fence(nb);
#ifdef CH_STAMP
        if (stage_stamps && (threadIdx.x & 63) == 0) stage_stamps[ks] = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) b[mt] = nb[mt];
    }
    ws.pos = base + NST;
}
// linear1 chunk: a1[nt][mt] (columns 32 wave + 16 nt + ..) += act[64 x 512] * W1 chunk; a stage = 2 k-steps of the wave's
// 32 rows: fragments [k-step 2][n-tile 2]
template <int MT, int NT = 4>
DEVINL void phase_ff1(f32x4_t (&a1)[NT / 2][MT], const char* abuf, WStreamT<NT>& ws, int lane) {
    constexpr int RD = ChW<NT>::D, NH = NT / 2;      // a stage = 2 k-steps x NH n-tiles of the wave's 16 NH chunk rows
    lane = fresh_v(lane);
    const FragOff fo = frag_off(abuf, lane);
    u32x4 b[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) b[mt] = frag_rd(fo, 0, mt);
    const unsigned base = ws.pos;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
        const int i = st % RD;
        u32x4 wk[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wk[nt] = ws.w[i][nt];
        ws_load(ws, i, base + st + RD);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const int ks = 2 * st + k2;
            u32x4 nb[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) nb[mt] = b[mt];
            if (ks + 1 < 16) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) nb[mt] = frag_rd(fo, ks + 1, mt);
            }
            __builtin_amdgcn_sched_barrier(0);   // see phase_n512
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
                for (int n = 0; n < NH; ++n) mma16(a1[n][mt], wk[NH * k2 + n], b[mt]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NH; n += 2) {
                if constexpr (NT == 8 && MT == 4)
                    asm volatile("" : "+a"(a1[n][0]), "+a"(a1[n][1]), "+a"(a1[n][2]), "+a"(a1[n][3]), "+a"(a1[n + 1][0]),
                                 "+a"(a1[n + 1][1]), "+a"(a1[n + 1][2]), "+a"(a1[n + 1][3]));
                else if constexpr (MT == 4)
                    asm volatile("" : "+v"(a1[n][0]), "+v"(a1[n][1]), "+v"(a1[n][2]), "+v"(a1[n][3]), "+v"(a1[n + 1][0]),
                                 "+v"(a1[n + 1][1]), "+v"(a1[n + 1][2]), "+v"(a1[n + 1][3]));
                else if constexpr (MT == 2)
                    asm volatile("" : "+v"(a1[n][0]), "+v"(a1[n][1]), "+v"(a1[n + 1][0]), "+v"(a1[n + 1][1]));
                else
                    asm volatile("" : "+v"(a1[n][0]), "+v"(a1[n + 1][0]));
            }
            frag_fence(nb);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) b[mt] = nb[mt];
        }
    }
    ws.pos = base + 8;
}

// LayerNorm statistics of the 64 rows over all 512 columns: this lane's rows are 16 mt + c.  One exchange: every wave
// publishes (sum, sum of squares) of its 64 columns, var = E[v^2] - mean^2 in fp32 (|mean| is of the order of the
// standard deviation for these activations: the cancellation costs ~1e-7 relative, far below the bf16 operands).
// Returns rstd and nmr = -mean * rstd: the normalised value is fma(v, rstd, nmr), one op per element instead of two.
// Lane l finishes row l of the block (reduce-scatter over the lane groups, then the 8 waves' pairs in wave order: the
// sums are deterministic), and the four lane groups exchange their rows' (rstd, nmr) by swaps.
template <int MT, int NT>
DEVINL void row_stats(const f32x4_t (&acc)[NT][MT], float* scr, int wave, int lane, float eps, float (&nmr)[MT], float (&rstd)[MT]) {
    lane = fresh_v(lane);
    wave = fresh_s(wave);
    float s[MT], s2[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x2_t t = {0.0f, 0.0f}, t2 = {0.0f, 0.0f};
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
                const f32x2_t v = {acc[nt][mt][q], acc[nt][mt][q + 1]};
                t += v;
                t2 = __builtin_elementwise_fma(v, v, t2);
            }
        s[mt] = t[0] + t[1];
        s2[mt] = t2[0] + t2[1];
    }
    // lane -> the row of the block it finishes: MT = 4: row `lane` (row tile g); MT = 2: row 16 (g & 1) + c; MT = 1: row c
    f32x2_t mine;
    int row;
    if constexpr (MT == 4) {
        mine[0] = rs4_sum(s[0], s[1], s[2], s[3]);
        mine[1] = rs4_sum(s2[0], s2[1], s2[2], s2[3]);
        row = lane;
    } else if constexpr (MT == 2) {
        float a0 = s[0], a1 = s[1], b0 = s2[0], b1 = s2[1];
        swap16(a0, a1);              // a0 = [s0.g0, s1.g0, s0.g2, s1.g2], a1 = [s0.g1, s1.g1, s0.g3, s1.g3]
        swap16(b0, b1);
        float p = a0 + a1, q = b0 + b1, p2 = p, q2 = q;          // p = [s0(g0+g1), s1(g0+g1), s0(g2+g3), s1(g2+g3)]
        swap32(p, p2);               // p = lower half everywhere, p2 = upper half everywhere
        swap32(q, q2);
        mine[0] = p + p2;            // lane groups 0, 2: row tile 0; 1, 3: row tile 1
        mine[1] = q + q2;
        row = (lane & 15) + 16 * ((lane >> 4) & 1);
    } else {
        mine[0] = ar4_sum(s[0]);
        mine[1] = ar4_sum(s2[0]);
        row = lane & 15;
    }
    *reinterpret_cast<f32x2_t*>(scr + (wave * 64 + row) * 2) = mine;       // (MT < 4: lane groups write the same value twice / 4 x)
    lds_barrier();
    f32x2_t tot = {0.0f, 0.0f};
#pragma unroll
    for (int w = 0; w < ChW<NT>::NW; ++w) tot += *reinterpret_cast<const f32x2_t*>(scr + (w * 64 + row) * 2);
    const float mean = tot[0] * (1.0f / 512.0f);
    const float var = fmaxf(tot[1] * (1.0f / 512.0f) - mean * mean, 0.0f);
    const float rs = rsqrtf(var + eps);
    const float nm = -mean * rs;
    if constexpr (MT == 4) {
        ag4(rs, rstd);
        ag4(nm, nmr);
    } else if constexpr (MT == 2) {
        float r0 = rs, r1 = rs, n0 = nm, n1 = nm;
        swap16(r0, r1);              // r0 = row tile 0's value in every lane group, r1 = row tile 1's
        swap16(n0, n1);
        rstd[0] = r0; rstd[1] = r1; nmr[0] = n0; nmr[1] = n1;
    } else {
        rstd[0] = rs;
        nmr[0] = nm;
    }
}

DEVINL f32x4_t lds4b(const char* base, int byte_off) {
    return *reinterpret_cast<const f32x4_t*>(base + byte_off);
}
DEVINL f32x4_t lds4(const char* base, int float_index) {
    return *reinterpret_cast<const f32x4_t*>(base + float_index * 4);
}
// Byte offset of this lane's first column (64 wave + 4 g) in a 512-float LDS vector, as a value the compiler cannot take
// apart: the per-iteration column offsets (64 nt bytes) then fold into the ds_read immediates.
template <int NT = 4>
DEVINL int col_base_bytes(int wave, int g) {
    int v = (16 * NT * wave + 4 * g) * 4;
    asm volatile("" : "+v"(v));
    return v;
}

// This lane's four rows of a [rows, 512] fp32 matrix (residual stream, rotary table), 4 column quads each (one per
// n-tile): a register pipeline 2 n-tiles deep (32 VGPRs).
// Layouts.  Row-major [row][512]: a load instruction then touches 16 rows x 64 bytes.  COLUMN-BLOCKED
// [64 groups of 8 columns][rows][8 floats]: the 16 rows of two lane groups are consecutive, so an instruction reads two
// contiguous half kilobytes.  The residual stream between chain launches and the rotary table handed to them are
// column-blocked; only layer 0's input when written by gemm_rowln is row-major (`xres_rowmajor`).
template <int MT>
struct RowPipe {
    __amdgpu_buffer_rsrc_t rsrc;   // the matrix as a raw buffer: address = SGPR descriptor + SGPR (n-tile) + VGPR (row, lane group)
    unsigned voff[MT];             // byte offset of this lane's 16 bytes of row tile mt inside the wave's first column group
    unsigned soff, its;            // byte offset of the wave's first column group, bytes per n-tile step (wave-uniform)
    f32x4_t q[2][MT];              // [n-tile & 1][row tile]
};
DEVINL __amdgpu_buffer_rsrc_t f32_buffer(const float* base, long n_floats) {
    const long bytes = n_floats * 4;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes < 0xFFFFFFFFl ? (int)bytes : -1, 0x00020000);
}
template <int MT>
DEVINL void rp_issue(RowPipe<MT>& rp, int nt) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        rp.q[nt & 1][mt] = __builtin_bit_cast(
            f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rp.rsrc, rp.voff[mt], rp.soff + (unsigned)nt * rp.its, 0));
}
// rows: row count of the column-blocked matrix, or 0 for a row-major one (`total_rows` rows of 512 floats)
template <int NT, int MT>
DEVINL void rp_start(RowPipe<MT>& rp, const float* base, const int (&row)[MT], long rows, long total_rows, int wave, int g) {
    rp.rsrc = f32_buffer(base, total_rows * 512);
    const unsigned grp = rows > 0 ? (unsigned)rows * 32u : 0u;       // bytes per column group of 8
    rp.its = rows > 0 ? 2u * grp : 64u;
    rp.soff = rows > 0 ? (unsigned)wave * (2u * NT) * grp : (unsigned)wave * (64u * NT);
    const unsigned gterm = rows > 0 ? (unsigned)(g >> 1) * grp + 16u * (g & 1) : 16u * g;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
        rp.voff[mt] = (rows > 0 ? (unsigned)row[mt] * 32u : (unsigned)row[mt] * 2048u) + gterm;
    rp_issue(rp, 0);
    rp_issue(rp, 1);
}
// store this lane's 16 bytes (columns 64 wave + 16 nt + 4 g ..) of row `row` of a column-blocked matrix of `rows` rows
template <int NT = 4>
DEVINL void cb_store(__amdgpu_buffer_rsrc_t rsrc, long rows, int wave, int nt, int row, int g, f32x4_t v) {
    const unsigned grp = (unsigned)rows * 32u;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc,
                                           (unsigned)row * 32u + (unsigned)(g >> 1) * grp + 16u * (g & 1),
                                           (unsigned)(wave * 2 * NT + 2 * nt) * grp, 0);
}

// LDS byte offset of this lane's 8 bytes (4 bf16: columns 16 nt + 4 g .. of the wave's k-tile) of row 16 mt + c in a
// [64][128 B] activation tile: + mt * 2048; the chunk is 2 nt + (g >> 1) (+ 4 for the odd half of a 32-column owner)
DEVINL int act_wr_off(int lane, int nt, int chunk0 = 0) {
    const int c = lane & 15, g = lane >> 4;
    return c * TC_ROWB + (((chunk0 + 2 * nt + (g >> 1)) ^ tile_swz(c)) << 4) + 8 * (g & 1);
}

// u = LayerNorm(acc) (optionally rotated) -> bf16 -> activation block in LDS (k = column); gv, bv: LDS vectors;
// rp: the rotary rows (cos0 sin0 cos1 sin1 per column quad), started by the caller before the statistics exchange
template <bool ROT, int MT, int NT>
DEVINL void norm_to_lds(const f32x4_t (&acc)[NT][MT], const float (&nmr)[MT], const float (&rstd)[MT], const char* gv,
                        const char* bv, RowPipe<MT>& rp, char* abuf, int wave, int lane, char* plain) {
    lane = fresh_v(lane);
    wave = fresh_s(wave);
    const int g = lane >> 4;
    const int cb0 = col_base_bytes<NT>(wave, g);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const f32x4_t g4 = lds4b(gv + cb0, 64 * nt), b4 = lds4b(bv + cb0, 64 * nt);
        const int wo = ((wave * NT + nt) >> 2) * 8192 + act_wr_off(lane, nt & 3);      // k-tile = 64 columns
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float u[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) u[t] = fmaf(fmaf(acc[nt][mt][t], rstd[mt], nmr[mt]), g4[t], b4[t]);
            if (plain) {   // the un-rotated image too (V = norm1(x) W_v)
                uint2 pk;
                pk.x = pack_bf2(u[0], u[1]);
                pk.y = pack_bf2(u[2], u[3]);
                *reinterpret_cast<uint2*>(plain + wo + mt * 2048) = pk;
            }
            if (ROT) {
                const f32x4_t q = rp.q[nt & 1][mt];   // cos0 sin0 cos1 sin1
                const float y0 = u[0] * q[0] - u[1] * q[1], y1 = u[1] * q[0] + u[0] * q[1];
                const float y2 = u[2] * q[2] - u[3] * q[3], y3 = u[3] * q[2] + u[2] * q[3];
                u[0] = y0; u[1] = y1; u[2] = y2; u[3] = y3;
            }
            uint2 pk;
            pk.x = pack_bf2(u[0], u[1]);
            pk.y = pack_bf2(u[2], u[3]);
            *reinterpret_cast<uint2*>(abuf + wo + mt * 2048) = pk;
        }
#ifndef CH_ABLATE_ROWLAT
        if (ROT && nt + 2 < NT) rp_issue(rp, nt + 2);
#endif
        CH_FENCE();   // one n-tile at a time (see the fc epilogue)
    }
}

// head-major scatter of a 512-wide projection (wave = head): model/model.py:78-80,92-95.  The accumulator layout gives a
// lane 8 bytes of a row at a time; written like that every store instruction makes 16 thirty-two-byte write requests.
// Instead each 32-row half of the wave's [64 rows][64 columns] tile goes through 4 KB of the (by now idle) constants area
// -- wave-private, XOR-swizzled by (row >> 1) & 7 (the 64 banks hold two 128-byte rows), no barrier -- and leaves as 16
// bytes per lane, 8 lanes per 128-byte row: 8 full lines per instruction.
DEVINL char* stage_area(char* smem, int wave) { return smem + (wave < 7 ? wave * 4096 : CH_STG7); }
template <bool SCALE, int MT, int NT>   // Q carries 1 / sqrt(d_k); K and V are stored as they are
DEVINL void store_heads(const f32x4_t (&acc)[NT][MT], void* base, float scale, int L, int Lp, int H, int m0, int M,
                        int wave, int lane, char* smem, int dn = 1, int dancer = 0) {
    // rows are FRAMES m0 .. of dancer `dancer` (token = frame dn + dancer; dn = 1: rows are tokens); L tokens per sequence
#ifdef CH_ABLATE_STORES   // timing experiment only: how much of the Q / K / V tail is the head-major scatter?
    if (M > 0) return;
#endif
    lane = fresh_v(lane);
    wave = fresh_s(wave);
    const int c = lane & 15, g = lane >> 4;
    char* stg = stage_area(smem, wave);
    const int row0 = lane >> 3, ch = lane & 7;      // read side: row row0 + 8 k, 16-byte chunk ch
    const int Lf = L / dn;                          // frames per sequence
    constexpr int NH = MT == 4 ? 2 : 1, NML = MT == 1 ? 1 : 2;      // 32-row halves of the block, row tiles per half
#pragma unroll
    for (int hh = 0; hh < NT / 4; ++hh) {            // the wave's heads (NT = 4: wave = head; NT = 8: two heads per wave)
    const int head = wave * (NT / 4) + hh;
#pragma unroll
    for (int hf = 0; hf < NH; ++hf) {
#pragma unroll
        for (int ml = 0; ml < NML; ++ml) {
            const int rl = 16 * ml + c;             // row of the 32-row staging tile
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4_t v = acc[4 * hh + nt][2 * hf + ml];
                uint2 pk;
                if (SCALE) {
                    pk.x = pack_bf2(v[0] * scale, v[1] * scale);
                    pk.y = pack_bf2(v[2] * scale, v[3] * scale);
                } else {
                    pk.x = pack_bf2(v[0], v[1]);
                    pk.y = pack_bf2(v[2], v[3]);
                }
                *reinterpret_cast<uint2*>(stg + rl * 128 + (((2 * nt + (g >> 1)) ^ ((rl >> 1) & 7)) << 4) + 8 * (g & 1)) = pk;
            }
        }
        // the LDS queue of a wave is in order: its reads below see its writes above
        int m = m0 + 32 * hf + row0;
        int seq;
        if (dn == 1) {                 // rows are tokens: the block starts in sequence m0 / L (scalar) and crosses at most once
            const int sq0 = m0 / L;
            seq = m >= (sq0 + 1) * L ? sq0 + 1 : sq0;
        } else {
            seq = m / Lf;
        }
        int tokf = m - seq * Lf;
        // destination of (sequence, head = wave, token, chunk): +8 frames = +8 dn tokens of 128 bytes; past the end of a
        // sequence the next one starts (H * Lp - L) rows further
        uint16_t* dst = reinterpret_cast<uint16_t*>(base) + (((long)seq * H + head) * Lp + tokf * dn + dancer) * 64 + ch * 8;
        const long wrap = ((long)H * Lp - L) * 64;
#pragma unroll
        for (int k = 0; k < 2 * NML; ++k) {
            const int row = row0 + 8 * k;
            const u32x4 v = *reinterpret_cast<const u32x4*>(stg + row * 128 + ((ch ^ ((row >> 1) & 7)) << 4));
            if (m < M) *reinterpret_cast<u32x4*>(dst) = v;
            m += 8;
            tokf += 8;
            dst += 8 * 64 * dn;
            if (tokf >= Lf) {
                tokf -= Lf;
                dst += wrap;
            }
        }
    }
    }
}

// ---- shared by chain.hip and chain_split.hip: the in-kernel attention's constants and lane maxima, the fragment-order stores ----
#define CH_LOG2E 1.4426950408889634f
#ifndef CH_ATT_THR
#define CH_ATT_THR 5.0f      // in-kernel attention: a tile moves the running maximum when a score exceeds it by more than this (log2 units)
#endif

// The lane's maximum over its 8 scores of a 32-key tile, not below `floor`.  Four v_max3_f32: written as nested 3-input maxima of
// fmaximum_num hipcc selects v_max3_f32 with no operand canonicalisation (fmaxf costs one more instruction per value, and the loops
// below are bound by the number of instructions a wave can issue).  NOT inline assembly (rounds 4-5: `asm("v_max3_f32 ..")`): an MFMA's
// result may only be read by a VALU instruction a number of wait states after the MFMA, the hardware does not interlock, and the
// compiler's hazard pass counts them for instructions it knows -- it cannot for an asm statement (it emitted `s_nop 7` in front of a
// builtin maximum of an accumulator and nothing in front of the asm one).  Round 5's loop read the scores a dozen MFMAs after they were
// written; a software-pipelined form of the loop whose prologue reads them at once got stale registers -- rows of wrong maxima, NaNs
// (profiles/r06_attention_pipeline.txt; the pipelined form itself is profiles/r06_attention_pipeline_experiment.patch).
DEVINL float max3n(float a, float b, float c) { return __builtin_fmaximum_numf(__builtin_fmaximum_numf(a, b), c); }
DEVINL float lane_max8(const f32x4_t& a, const f32x4_t& b, float floor) {
    return max3n(max3n(max3n(max3n(a[0], a[1], a[2]), a[3], b[0]), b[1], b[2]), b[3], floor);
}

// The next layer's Q^T / K / V^T in the order the in-kernel attention loads them, straight from the accumulators (NT = 4: wave =
// head): every store is a 1-KB piece of one wave instruction.  The block is block `bis` of sequence `seq` (tcdiff_chain_args.seq_blocks).
// Q: private to the (block, wave) that reads it back in the next launch -- [lblk][wave][mt < 4][d-step s][lane][8], scaled by
// log2(e) / sqrt(d_k) (the attention works in the exp2 domain).
template <int MT>
DEVINL void store_qfrag(const f32x4_t (&acc)[4][MT], void* base, float scale, int lblk, int wave, int lane) {
    u32x4* dst = reinterpret_cast<u32x4*>(base) + ((long)(lblk * 8 + wave) * 8) * 64 + lane;     // 8 pieces per (block, wave) whatever MT
    scale *= CH_LOG2E;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4_t lo = acc[2 * s][mt], hi = acc[2 * s + 1][mt];
            u32x4 f;
            f[0] = pack_bf2(lo[0] * scale, lo[1] * scale);
            f[1] = pack_bf2(lo[2] * scale, lo[3] * scale);
            f[2] = pack_bf2(hi[0] * scale, hi[1] * scale);
            f[3] = pack_bf2(hi[2] * scale, hi[3] * scale);
            dst[(mt * 2 + s) * 64] = f;
        }
}
// K: the 16 keys of row tile mt and d-step s are piece 2 (first_row / 16 + mt) + s of the (sequence, head) image [nkt][4][64 lanes][8]
// (tcdiff_pack_kv_frags' K order: tile = 32 keys, piece = 2 (key half) + d-step); pieces of tiles >= nkt (rows past the last tile
// of the sequence: clamped copies) are dropped.
template <int MT>
DEVINL void store_kfrag(const f32x4_t (&acc)[4][MT], void* base, int seq, int first_row, int nkt, int wave, int lane) {
    u32x4* dst = reinterpret_cast<u32x4*>(base) + ((long)(seq * 8 + wave) * nkt) * 256 + lane;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int pc = 2 * ((first_row >> 4) + mt);
        if ((pc >> 2) < nkt) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f32x4_t lo = acc[2 * s][mt], hi = acc[2 * s + 1][mt];
                u32x4 f = {pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3]), pack_bf2(hi[0], hi[1]), pack_bf2(hi[2], hi[3])};
                dst[(pc + s) * 64] = f;
            }
        }
    }
}
// V: accT = the TRANSPOSED projection tiles (phase_n512<.., SWAP>): lane (c, g) holds keys 16 mt + 4 g + j of feature 16 nt + c,
// and the pair of row tiles (2 t, 2 t + 1) IS the A operand of O^T += V^T P^T for the 32-key tile (MT / 2) bis + t and d tile nt.
// first_row: the block's first row within its sequence.  MT = 1 (16-row blocks, or a sequence's last block when it holds <= 16
// rows): the block owns one half of every lane's 16 bytes -- keys 16 (first_row / 16 & 1) .. + 15 of the tile; a half that lies past
// the sequence is never written (the caller's image starts zeroed and only ever holds finite values; those keys are masked).
template <int MT>
DEVINL void store_vfrag(const f32x4_t (&accT)[4][MT], void* base, int seq, int first_row, int nkt, int Lseq, int wave, int lane) {
    u32x4* dst = reinterpret_cast<u32x4*>(base) + ((long)(seq * 8 + wave) * nkt) * 256 + lane;
    if constexpr (MT == 1) {
        const int kt = first_row >> 5, half = (first_row >> 4) & 1;
        if (kt < nkt) {
            // the sequence's last row tile in the FIRST half of its 32-key tile: nobody owns the second half -- those keys are masked,
            // but a masked P = 0 times a non-finite V is NaN, so the half is written (zeros) rather than left to the allocation
            const bool tail = half == 0 && first_row + 16 >= Lseq;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4_t lo = accT[nt][0];
                uint2 f = {pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3])};
                char* d8 = reinterpret_cast<char*>(dst + (kt * 4 + nt) * 64);
                *reinterpret_cast<uint2*>(d8 + 8 * half) = f;
                if (tail) *reinterpret_cast<uint2*>(d8 + 8) = uint2{0u, 0u};
            }
        }
        return;
    }
    const int bis = first_row / (16 * MT);
#pragma unroll
    for (int t = 0; t < MT / 2; ++t) {
        const int kt = (MT / 2) * bis + t;
        if (kt < nkt) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const f32x4_t lo = accT[nt][2 * t], hi = accT[nt][2 * t + 1];
                u32x4 f = {pack_bf2(lo[0], lo[1]), pack_bf2(lo[2], lo[3]), pack_bf2(hi[0], hi[1]), pack_bf2(hi[2], hi[3])};
                dst[(kt * 4 + nt) * 64] = f;
            }
        }
    }
}

