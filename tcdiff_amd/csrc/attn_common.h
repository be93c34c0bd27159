// Shared by attention.hip (inference) and attention_train.hip (train-mode forward + backward): the staged K / V tile
// layout and the MFMA fragment reads of common.h's two arithmetic policies.
#pragma once
#include "common.h"
#include <type_traits>

typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

template <class P>
struct AttnCfg {
    static constexpr int ES = sizeof(typename P::elem_t);
    static constexpr int KB = P::KT;                 // keys per staged tile
    static constexpr int NKT = KB / 32;              // 32-key MFMA tiles per staged tile
    static constexpr int DSUB = 64 * ES / TC_ROWB;   // 128-B sub-tiles covering d = 0..63 of a row (bf16 1, f32 2)
    static constexpr int NKS = 64 * ES / 32;         // k-steps over d for S^T
    static constexpr int PV_STEPS = ES;              // k-steps over one 32-key tile for O^T: bf16 2 (16 keys), f32 4 (8 keys)
    static constexpr int TILE_BYTES = KB * 64 * ES;  // 8 KB
    static constexpr int STAGE = 2 * TILE_BYTES;     // [K tile | V tile]
};

// byte offset of element (row, d) inside a staged [KB][64] tile (sub-tiles of 128-B rows, swizzled chunks)
template <class P>
DEVINL int kv_off(int row, int d) {
    constexpr int ES = sizeof(typename P::elem_t);
    constexpr int EPR = TC_ROWB / ES;   // elements per 128-B sub-row: bf16 64, f32 32
    constexpr int EPC = 16 / ES;
    const int sub = d / EPR, dd = d % EPR;
    return sub * (P::KT * TC_ROWB) + tile_off(row, dd / EPC) + (dd % EPC) * ES;
}

// A operand of O^T += V^T P^T for output features d = dt*32 + (lane & 31) and the keys of k-step `st` of the
// 32-key tile `kt` (element order matches the P^T accumulator registers, see attention_kernel).
template <class P>
DEVINL u32x4 v_frag(const char* vt, int dt, int kt, int st, int lane) {
    if (P::IS_BF16) {
        // element j <-> key kt*32 + 16*st + 8*(j>>2) + 4*h + (j&3): two transposed reads of 4 keys x 16 features
        const int h = lane >> 5, i16 = lane & 15, q = i16 >> 2, p = i16 & 3, g1 = (lane >> 4) & 1;
        const int kb = kt * 32 + 16 * st + 4 * h;
        const int d = dt * 32 + 16 * g1 + 4 * p;
        typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vt + kv_off<P>(kb + q, d)));
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vt + kv_off<P>(kb + 8 + q, d)));
        const u32x2_t l2 = __builtin_bit_cast(u32x2_t, lo), h2 = __builtin_bit_cast(u32x2_t, hi);
        u32x4 out = {l2[0], l2[1], h2[0], h2[1]};
        return out;
    } else if constexpr (P::IS_X3) {
        // bf16x3: element (key, d) of the staged V tile = bf16 slots (d & 3) [hi] and 4 + (d & 3) [lo] of the 16-byte chunk d / 4 of
        // the key's row.  The fragment wants keys k0 .. k0 + 3 of ONE d: hi quad | lo quad, gathered with 2-byte reads.
        const int r = lane & 31, h = lane >> 5;
        const int d = dt * 32 + r;
        const int k0 = kt * 32 + 8 * st + 4 * h;
        uint32_t hv[4], lv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const char* c = vt + kv_off<P>(k0 + j, d & ~3) + 2 * (d & 3);
            hv[j] = *reinterpret_cast<const uint16_t*>(c);
            lv[j] = *reinterpret_cast<const uint16_t*>(c + 8);
        }
        const u32x4 out = {hv[0] | (hv[1] << 16), hv[2] | (hv[3] << 16), lv[0] | (lv[1] << 16), lv[2] | (lv[3] << 16)};
        return out;
    } else {
        // MFMA j of the k-step takes key kt*32 + 8*st + 4*h + j
        const int r = lane & 31, h = lane >> 5;
        const int d = dt * 32 + r;
        const int k0 = kt * 32 + 8 * st + 4 * h;
        u32x4 out;
        out[0] = *reinterpret_cast<const uint32_t*>(vt + kv_off<P>(k0 + 0, d));
        out[1] = *reinterpret_cast<const uint32_t*>(vt + kv_off<P>(k0 + 1, d));
        out[2] = *reinterpret_cast<const uint32_t*>(vt + kv_off<P>(k0 + 2, d));
        out[3] = *reinterpret_cast<const uint32_t*>(vt + kv_off<P>(k0 + 3, d));
        return out;
    }
}

