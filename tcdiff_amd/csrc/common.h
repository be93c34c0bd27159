// Common device helpers for the TCDiff gfx950 (MI355X / CDNA4) kernels.
//
// Two arithmetic policies share every kernel template:
//   * MmaBF16 : operands bf16 in HBM/LDS, v_mfma_f32_32x32x16_bf16, fp32 accumulate  (throughput mode)
//   * MmaF32  : operands fp32,             v_mfma_f32_32x32x2_f32  (exact fp32 fma chain; parity mode)
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
DEVINL uint32_t pack_bf2(float lo, float hi) { return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16); }

struct MmaBF16 {
    typedef uint16_t elem_t;
    static constexpr int KT = 64;        // elements of K per 128-B staged row
    static constexpr int EPC = 8;        // elements per 16-B chunk
    static constexpr bool IS_BF16 = true;
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

// Row of the C/D tile held in accumulator register `reg` by a lane of half `h`.
DEVINL int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// ---- staged-tile addressing -------------------------------------------------------------------
// A staged tile is [rows][8 chunks of 16 B]; chunk c of row r lives at chunk slot c ^ ((r >> 1) & 7).
// ds_read_b128 serves 16-lane groups {0-3,12-15,20-27}/{4-11,16-19,28-31} (+32) against a 256-B bank
// row = two 128-B tile rows.  A fragment read has lane -> row, same chunk index: with this XOR every
// group touches the 8 slot values {c^0..c^7} once per row parity = 16 distinct 16-B slots: conflict-free
// (MI355X_MICROARCH.md LDS table; the plain (r & 7) form is 2-way).  ds_write_b128 serves 8 contiguous
// lanes = one row's 8 chunks: conflict-free for any per-row permutation.
DEVINL int tile_off(int row, int chunk) { return row * TC_ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

// ---- activations ---------------------------------------------------------------------------------
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, i.e. fp32 rounding level): 1 exp + 1 rcp + 6 fma instead of
// libm's ~50-instruction erff.  GELU keeps the reference's exact-erf definition (F.gelu default, TCDiff.py:85).
DEVINL float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float y = 1.0f - p * t * __expf(-ax * ax);
    return copysignf(y, x);
}
DEVINL float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
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
DEVINL float apply_act(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.0f);
        case ACT_GELU: return gelu_erf(v);
        case ACT_MISH: return mish_f(v);
        case ACT_SILU: return silu_f(v);
        default: return v;
    }
}

// ---- wave reductions over the 32 lanes of a half-wave (xor masks < 32 never cross halves) -------
DEVINL float half_sum(float v) {
    v += __shfl_xor(v, 1);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 16);
    return v;
}
DEVINL float wave_sum(float v) {
    v = half_sum(v);
    v += __shfl_xor(v, 32);
    return v;
}
DEVINL float wave_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 1));
    v = fmaxf(v, __shfl_xor(v, 2));
    v = fmaxf(v, __shfl_xor(v, 4));
    v = fmaxf(v, __shfl_xor(v, 8));
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 32));
    return v;
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

// error codes of the C ABI
#define TC_OK 0
#define TC_ERR_ARG (-1)
#define TC_ERR_ALIGN (-2)
#define TC_ERR_LAUNCH (-3)
#define TC_ERR_UNSUPPORTED (-4)

#define TC_CHECK_LAUNCH()                                   \
    do {                                                    \
        hipError_t e_ = hipGetLastError();                  \
        if (e_ != hipSuccess) return TC_ERR_LAUNCH;         \
    } while (0)
