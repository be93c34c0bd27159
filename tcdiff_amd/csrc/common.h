// Common device helpers for the TCDiff gfx950 (MI355X / CDNA4) kernels.
//
// Three arithmetic policies share the kernel templates:
//   * MmaBF16   : operands bf16 in HBM/LDS, v_mfma_f32_32x32x16_bf16, fp32 accumulate  (throughput mode)
//   * MmaF32    : operands fp32,             v_mfma_f32_32x32x2_f32  (exact fp32 fma chain; parity mode)
//   * MmaBF16x3 : operands fp32 (MmaF32's storage), each product as three bf16 MFMAs on (hi, lo) splits (fast parity mode)
// Both policies stage operand tiles as rows of 128 bytes (64 bf16 / 32 f32 along K) and every lane
// fetches its MFMA fragment as ONE 16-byte LDS read, so staging, swizzle and addressing are
// byte-identical for the two; only the MFMA issue differs.
//
// Fragment convention (one "k-step" = 32 bytes of K per row = 2 chunks of 16 B):
//   lane l: r = l & 31 (tile row of A / tile column of B), h = l >> 5 (which 16-B chunk of the k-step).
//   bf16 : chunk holds k = 16*ks + 8*h + j, j=0..7      -> one 32x32x16 MFMA per k-step
//   f32  : chunk holds k =  8*ks + 4*h + j, j=0..3      -> four 32x32x2 MFMAs per k-step, MFMA j pairs
//          (h=0: k=8ks+j) with (h=1: k=8ks+4+j).  The K order inside a dot product is permuted, the
//          same way for A and B, which only changes the fp32 summation order.
// C/D layout (both): col = l & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(l >> 5), reg in [0,16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;  // one 16-byte LDS/global access

#define TC_WAVE 64
#define TC_ROWB 128  // bytes of K per staged row
#define DEVINL __device__ __forceinline__

// ---- bf16 conversion (round-to-nearest-even; the plain cast lowers to v_cvt_pk_bf16_f32) -------
DEVINL uint16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(uint16_t, b);
}
DEVINL float bf2f(uint16_t u) { return __builtin_bit_cast(float, (uint32_t)u << 16); }
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// two floats -> one dword of two bf16 (lo in bits 0..15): a single v_cvt_pk_bf16_f32, no shift/or
DEVINL uint32_t pack_bf2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

struct MmaBF16 {
    typedef uint16_t elem_t;
    static constexpr int KT = 64;        // elements of K per 128-B staged row
    static constexpr int EPC = 8;        // elements per 16-B chunk
    static constexpr bool IS_BF16 = true;
    static constexpr bool IS_X3 = false;
    // (the 4-element chunk helpers of the 4-byte policies: referenced from the untaken side of `if (P::IS_BF16)` branches only)
    static DEVINL u32x4 chunk_from4(const f32x4_t& f) { return __builtin_bit_cast(u32x4, f); }
    static DEVINL f32x4_t chunk_to4(const u32x4& c) { return __builtin_bit_cast(f32x4_t, c); }
    static DEVINL elem_t from_f32(float f) { return f2bf(f); }
    static DEVINL float to_f32(elem_t e) { return bf2f(e); }
    static DEVINL void mma(f32x16_t& acc, const u32x4& a, const u32x4& b) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a),
                                                      __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
    }
};

struct MmaF32 {
    typedef float elem_t;
    static constexpr int KT = 32;
    static constexpr int EPC = 4;
    static constexpr bool IS_BF16 = false;
    static constexpr bool IS_X3 = false;
    static DEVINL u32x4 chunk_from4(const f32x4_t& f) { return __builtin_bit_cast(u32x4, f); }   // four elements <-> a 16-byte chunk
    static DEVINL f32x4_t chunk_to4(const u32x4& c) { return __builtin_bit_cast(f32x4_t, c); }
    static DEVINL elem_t from_f32(float f) { return f; }
    static DEVINL float to_f32(elem_t e) { return e; }
    static DEVINL void mma(f32x16_t& acc, const u32x4& a, const u32x4& b) {
        // NOTE: bit-cast the WHOLE vector, then index.  __builtin_bit_cast(float, a.y) on a single element of an
        // ext_vector is miscompiled by hipcc 7.2 (every element reads lane register .x).
        const f32x4_t af = __builtin_bit_cast(f32x4_t, a), bf = __builtin_bit_cast(f32x4_t, b);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], acc, 0, 0, 0);
    }
};

// Split-bf16 ("bf16x3"): every product a b is evaluated as a_hi b_hi + a_hi b_lo + a_lo b_hi with a_hi = bf16(a), a_lo = bf16(a - a_hi)
// (lo lo, ~2^-18 of the product, is dropped): three bf16 MFMAs with fp32 accumulation instead of four exact-fp32 ones at 1/16 of
// the rate.  Each operand keeps ~16 significant bits, so a product is good to ~2^-16 relative -- two orders of magnitude inside
// the north-star's 1e-3 over the whole sampler (tests/test_parity_gpu.py, compute_dtype = "bf16x3").
// STORAGE: an element still takes 4 bytes at the position fp32 would give it (so every size, stride, staging and swizzle rule of
// MmaF32 applies unchanged), but a 16-byte chunk of four consecutive elements e0..e3 is kept ALREADY SPLIT,
//     [hi(e0) hi(e1) hi(e2) hi(e3) | lo(e0) lo(e1) lo(e2) lo(e3)]   (eight bf16),
// by whoever produces it (GEMM / attention / elementwise epilogues, the host's weight packer): the 16 bytes a lane fetches as
// its MFMA fragment ARE the A / B operands of v_mfma_f32_32x32x8_bf16_1k (lane half h supplies k = 4 h .. 4 h + 3) -- the first
// 8 bytes the hi quad, the second the lo quad -- and the k-loops contain no conversion at all.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
struct x3_word { uint32_t u; };      // a 4-byte storage slot of the split format: never a value by itself (chunk_to4 / load_T)
struct MmaBF16x3 {
    typedef x3_word elem_t;
    static constexpr int KT = 32;
    static constexpr int EPC = 4;
    static constexpr bool IS_BF16 = false;
    static constexpr bool IS_X3 = true;
    static DEVINL u32x4 chunk_from4(const f32x4_t& f) {
        const uint32_t h01 = pack_bf2(f[0], f[1]), h23 = pack_bf2(f[2], f[3]);      // round to nearest even
        const float g0 = __builtin_bit_cast(float, h01 << 16), g1 = __builtin_bit_cast(float, h01 & 0xFFFF0000u);
        const float g2 = __builtin_bit_cast(float, h23 << 16), g3 = __builtin_bit_cast(float, h23 & 0xFFFF0000u);
        const u32x4 c = {h01, h23, pack_bf2(f[0] - g0, f[1] - g1), pack_bf2(f[2] - g2, f[3] - g3)};   // the differences are exact
        return c;
    }
    static DEVINL f32x4_t chunk_to4(const u32x4& c) {
        const f32x4_t f = {__builtin_bit_cast(float, c[0] << 16) + __builtin_bit_cast(float, c[2] << 16),
                           __builtin_bit_cast(float, c[0] & 0xFFFF0000u) + __builtin_bit_cast(float, c[2] & 0xFFFF0000u),
                           __builtin_bit_cast(float, c[1] << 16) + __builtin_bit_cast(float, c[3] << 16),
                           __builtin_bit_cast(float, c[1] & 0xFFFF0000u) + __builtin_bit_cast(float, c[3] & 0xFFFF0000u)};
        return f;
    }
    // TWO consecutive k-steps at once (K = 16): the hi quads of the two chunks side by side are the 8-element operand of
    // v_mfma_f32_32x32x16_bf16 (lane half h supplies k = 8 h + j: j < 4 from the first k-step's chunk, j >= 4 from the second's;
    // A and B alike), which costs the same 32 cycles as the K = 8 form below: half the matrix-pipe time per product.  Every k-loop
    // of the split-bf16 kernels walks its k-steps in pairs; `mma` (one k-step) remains for odd tails.
    static DEVINL void mma2(f32x16_t& acc, const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1) {
        const u32x4 ah = {a0[0], a0[1], a1[0], a1[1]}, al = {a0[2], a0[3], a1[2], a1[3]};
        const u32x4 bh = {b0[0], b0[1], b1[0], b1[1]}, bl = {b0[2], b0[3], b1[2], b1[3]};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, al), __builtin_bit_cast(bf16x8_t, bh), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, bl), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, ah), __builtin_bit_cast(bf16x8_t, bh), acc, 0, 0, 0);
    }
    static DEVINL void mma(f32x16_t& acc, const u32x4& a, const u32x4& b) {
        const uint2 a_h = {a[0], a[1]}, a_l = {a[2], a[3]}, b_h = {b[0], b[1]}, b_l = {b[2], b[3]};
        const s16x4_t ah = __builtin_bit_cast(s16x4_t, a_h), al = __builtin_bit_cast(s16x4_t, a_l);
        const s16x4_t bh = __builtin_bit_cast(s16x4_t, b_h), bl = __builtin_bit_cast(s16x4_t, b_l);
        acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(al, bh, acc, 0, 0, 0);       // the small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ah, bh, acc, 0, 0, 0);
    }
};

// One T-typed element by index (the elementwise kernels of ops.hip: conversions, embeddings, small tables -- not hot paths)
template <class P>
DEVINL void store_T(typename P::elem_t* base, long i, float v) {
    if constexpr (std::is_same<P, MmaBF16x3>::value) {
        const uint16_t hi = f2bf(v);
        uint16_t* c = reinterpret_cast<uint16_t*>(base) + (i >> 2) * 8 + (i & 3);
        c[0] = hi;
        c[4] = f2bf(v - bf2f(hi));
    } else {
        base[i] = P::from_f32(v);
    }
}
template <class P>
DEVINL float load_T(const typename P::elem_t* base, long i) {
    if constexpr (std::is_same<P, MmaBF16x3>::value) {
        const uint16_t* c = reinterpret_cast<const uint16_t*>(base) + (i >> 2) * 8 + (i & 3);
        return bf2f(c[0]) + bf2f(c[4]);
    } else {
        return P::to_f32(base[i]);
    }
}

// Row of the C/D tile held in accumulator register `reg` by a lane of half `h`.
DEVINL int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// ---- staged-tile addressing -------------------------------------------------------------------
// A staged tile is [rows][8 chunks of 16 B]; chunk c of row r lives at chunk slot c ^ ((r >> 1) & 7).
// ds_read_b128 serves 16-lane groups {0-3,12-15,20-27}/{4-11,16-19,28-31} (+32) against a 256-B bank
// row = two 128-B tile rows.  A fragment read has lane -> row, same chunk index: with this XOR every
// group touches the 8 slot values {c^0..c^7} once per row parity = 16 distinct 16-B slots: conflict-free
// (MI355X_MICROARCH.md LDS table; the plain (r & 7) form is 2-way).  ds_write_b128 serves 8 contiguous
// lanes = one row's 8 chunks: conflict-free for any per-row permutation.
// Which bijection of (row >> 1) & 7 is used matters for the TRANSPOSED reads (ds_read_b64_tr_b16 of a V^T / K^T / Q^T
// fragment, attn_common.h v_frag): a 32-lane group of those touches 4 CONSECUTIVE rows x 64 contiguous bytes, and rows q
// and q + 2 (same 128-byte half of the bank row) must land in different 64-byte quarters -- with the plain value they
// differ in chunk bit 0 only and collide 2-way (round 2: SQ_LDS_BANK_CONFLICT = 34 % of the LDS-active cycles of
// attention_res, i.e. every transposed read took two passes).  Bit-reversing the three bits moves row bit 1 to chunk bit 2:
// the four rows then cover all 64 banks once.
DEVINL int tile_swz(int row) {
    const int x = row >> 1;
    return ((x & 1) << 2) | (x & 2) | ((x >> 2) & 1);
}
DEVINL int tile_off(int row, int chunk) { return row * TC_ROWB + ((chunk ^ tile_swz(row)) << 4); }

// ---- activations ---------------------------------------------------------------------------------
// GELU keeps the reference's exact-erf definition (F.gelu default, TCDiff.py:85) with erf by Abramowitz-Stegun 7.1.28,
//   erf(z) = 1 - 1 / (1 + a1 z + ... + a6 z^6)^16,  |error| <= 3e-7 for z >= 0,
// i.e. six FMAs, four squarings and ONE quarter-rate op (v_rcp_f32) per element instead of libm's ~50 instructions
// (7.1.26, used before, needs an exp as well).  With r = 1 / (...)^16 and erf odd,
//   gelu(x) = 0.5 x (1 + erf(x / sqrt 2)) = max(x, 0) - |0.5 x r|.
// The epilogue of linear1 evaluates 14.7 M of these per step and layer, more VALU time than the GEMM has MFMA time,
// so the two-element form below is written on float2 vectors for v_pk_fma_f32 / v_pk_mul_f32.
DEVINL float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    float p = fmaf(0.0000430638f, z, 0.0002765672f);
    p = fmaf(p, z, 0.0001520143f);
    p = fmaf(p, z, 0.0092705272f);
    p = fmaf(p, z, 0.0422820123f);
    p = fmaf(p, z, 0.0705230784f);
    p = fmaf(p, z, 1.0f);
    p = p * p; p = p * p; p = p * p; p = p * p;
    const float hr = 0.5f * x * __builtin_amdgcn_rcpf(p);   // v_rcp_f32 (1 ulp); __frcp_rn expands to an 11-instruction division
    return fmaxf(x, 0.0f) - fabsf(hr);
}
typedef float f32x2_t __attribute__((ext_vector_type(2)));
DEVINL f32x2_t gelu_erf2(f32x2_t x) {
    const f32x2_t ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2_t z = ax * 0.70710678118654752440f;
    f32x2_t p = __builtin_elementwise_fma((f32x2_t)(0.0000430638f), z, (f32x2_t)(0.0002765672f));
    p = __builtin_elementwise_fma(p, z, (f32x2_t)(0.0001520143f));
    p = __builtin_elementwise_fma(p, z, (f32x2_t)(0.0092705272f));
    p = __builtin_elementwise_fma(p, z, (f32x2_t)(0.0422820123f));
    p = __builtin_elementwise_fma(p, z, (f32x2_t)(0.0705230784f));
    p = __builtin_elementwise_fma(p, z, (f32x2_t)(1.0f));
    p = p * p; p = p * p; p = p * p; p = p * p;
    const f32x2_t r = {__builtin_amdgcn_rcpf(p[0]), __builtin_amdgcn_rcpf(p[1])};
    // max(x, 0) - |0.5 x r| = 0.5 (x + |x|) - 0.5 |x| r, all in packed ops (r > 0)
    const f32x2_t ha = ax * 0.5f;
    return __builtin_elementwise_fma(-ha, r, __builtin_elementwise_fma(x, (f32x2_t)(0.5f), ha));
}
DEVINL float softplus_t(float x) { return x > 20.0f ? x : log1pf(expf(x)); }  // torch threshold 20
DEVINL float mish_f(float x) { return x * tanhf(softplus_t(x)); }
DEVINL float silu_f(float x) { return x / (1.0f + expf(-x)); }

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_MISH = 3, ACT_SILU = 4 };
// Compile-time activation for the hot epilogues (a runtime switch gets if-converted: every lane then evaluates
// erf, tanh, log1p and exp for every element and selects -- measured 8.5 us per 128x128 tile).
template <int ACT>
DEVINL float act_ct(float v, int act_rt) {
    if (ACT == ACT_NONE) return v;
    if (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    if (ACT == ACT_GELU) return gelu_erf(v);
    return act_rt == ACT_MISH ? mish_f(v) : silu_f(v);   // ACT == 3: the two setup-only activations
}
// four consecutive outputs at once (GELU in its packed form)
template <int ACT>
DEVINL void act4_ct(float (&v)[4], int act_rt) {
    if (ACT == ACT_GELU) {
        const f32x2_t a = gelu_erf2(f32x2_t{v[0], v[1]}), b = gelu_erf2(f32x2_t{v[2], v[3]});
        v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = act_ct<ACT>(v[t], act_rt);
    }
}
DEVINL float apply_act(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.0f);
        case ACT_GELU: return gelu_erf(v);
        case ACT_MISH: return mish_f(v);
        case ACT_SILU: return silu_f(v);
        default: return v;
    }
}

// ---- wave reductions ---------------------------------------------------------------------------------
// __shfl_xor lowers to ds_bpermute_b32 (an LDS round trip, >100 cycles each, 6 dependent ones per wave-wide sum).
// These use DPP row operations for the 16-lane rows (4 VALU ops) and v_readlane for the 4 row totals.
template <int CTRL>
DEVINL float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
#define TC_DPP_QUAD_XOR1 0xB1     // quad_perm [1,0,3,2]
#define TC_DPP_QUAD_XOR2 0x4E     // quad_perm [2,3,0,1]
#define TC_DPP_ROW_HALF_MIRROR 0x141
#define TC_DPP_ROW_MIRROR 0x140
DEVINL float row16_sum(float v) {   // every lane ends with the sum over its 16-lane row
    v += dpp_mov<TC_DPP_QUAD_XOR1>(v);
    v += dpp_mov<TC_DPP_QUAD_XOR2>(v);
    v += dpp_mov<TC_DPP_ROW_HALF_MIRROR>(v);
    v += dpp_mov<TC_DPP_ROW_MIRROR>(v);
    return v;
}
DEVINL float lane_bcast(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
DEVINL float wave_sum(float v) {
    v = row16_sum(v);
    return (lane_bcast(v, 0) + lane_bcast(v, 16)) + (lane_bcast(v, 32) + lane_bcast(v, 48));
}
// value held by the lane 32 positions away (lane l <-> l ^ 32), via v_permlane32_swap
DEVINL float other_half(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    // lanes < 32: r[0] = own, r[1] = partner; lanes >= 32: r[0] = partner, r[1] = own
    const unsigned own_lo = r[0], own_hi = r[1];
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? own_lo : own_hi);
}

// ---- XCD-aware workgroup remap ---------------------------------------------------------------------
// Workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MB L2).  Remap the hardware block id
// so that every XCD works on a CONTIGUOUS range of logical tiles: tiles that share an operand panel (the column
// tiles of one row panel, the query blocks of one (sequence, head)) then hit the same L2.  Bijective for any
// block count (cdna_hip_programming.md T1); speed only, never correctness.
DEVINL int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, rem = nblocks & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < rem ? xcd * (q + 1) : rem * (q + 1) + (xcd - rem) * q) + idx;
}

// Barrier that also retires this wave's global->LDS DMA (global_load_lds): hipcc does NOT reliably put the
// s_waitcnt vmcnt(0) in front of __syncthreads() for LDS-DMA issued in an earlier basic block, so it is explicit.
DEVINL void sync_dma() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// ---- global -> LDS DMA staging (shared by gemm.hip and chain.hip) ----------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void gbl_void_t;

// One LDS-DMA instruction in its SGPR-base form: global_load_lds_dwordx4 voffset, s[base:base+1] with M0 = the LDS
// destination (one address VGPR instead of two; the main loops got 3-13 % shorter).  Written in asm because hipcc picks
// the 64-bit VGPR address form whenever the base moves inside a loop.  M0 is compiler-reserved, so it is saved and
// restored inside the statement (cdna_hip_programming.md section 5.7); completion is waited for by sync_dma().
DEVINL void glds16(const char* base, unsigned voff, const char* lds_dst) {
    const uintptr_t b = reinterpret_cast<uintptr_t>(base);
    const uint64_t sb = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)(b & 0xffffffffu));
    const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)reinterpret_cast<uintptr_t>(lds_dst));
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sb), "s"(dst) : "memory");
}

// Issue the global->LDS DMA of a [ROWS][128 B] tile.  `wave` must be wave-uniform.
// rows >= row_limit are clamped to row_limit-1 (their results are never stored).
template <int ROWS, int NWAVES>
DEVINL void stage_glds(char* lds_tile, const char* src, long ld_bytes, int row0, int row_limit, int row_mod, int wave,
                       int lane) {
    constexpr int PER = ROWS / 8 / NWAVES;
    static_assert(ROWS % (8 * NWAVES) == 0, "tile rows must divide over the waves");
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int blk = wave * PER + i;           // 1-KiB block = 8 tile rows
        const int row = blk * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ tile_swz(row);
        int gr = row0 + row;
        gr = gr < row_limit ? gr : row_limit - 1;
        if (row_mod > 0) gr = gr % row_mod;
        // 32-bit per-lane offset from the wave-uniform base (the launchers check that an operand spans < 4 GB)
        const unsigned off = (unsigned)gr * (unsigned)ld_bytes + (unsigned)(chunk * 16);
        glds16(src, off, lds_tile + blk * 1024);
    }
}

DEVINL u32x4 lds_frag(const char* tile, int row, int chunk) {
    return *reinterpret_cast<const u32x4*>(tile + tile_off(row, chunk));
}


// error codes of the C ABI
#define TC_OK 0
#define TC_ERR_ARG (-1)
#define TC_ERR_ALIGN (-2)
#define TC_ERR_LAUNCH (-3)
#define TC_ERR_UNSUPPORTED (-4)

// Per-device, thread-safe launcher state: the dynamic-LDS limit of a kernel has to be raised on EVERY device that
// launches it, and the CU count is a property of the device (not of the process).
#ifdef __cplusplus
#include <mutex>
#define TC_MAX_DEVICES 64
struct tc_dev_state {
    std::mutex mu;
    bool ready[TC_MAX_DEVICES] = {};
    int n_cu[TC_MAX_DEVICES] = {};
};
// Runs `setup(dev)` (returns hipSuccess on success) once per device; returns the device's CU count or a negative
// TC_ERR_* code.
template <class F>
static inline int tc_device_once(tc_dev_state& st, F setup) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TC_MAX_DEVICES) return TC_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(st.mu);
    if (!st.ready[dev]) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            return TC_ERR_UNSUPPORTED;
        if (setup(dev) != hipSuccess) return TC_ERR_UNSUPPORTED;
        st.n_cu[dev] = v;
        st.ready[dev] = true;
    }
    return st.n_cu[dev];
}
#endif

#define TC_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t e_ = hipGetLastError();                  \
        if (e_ != hipSuccess) return TC_ERR_LAUNCH;         \
    } while (0)
