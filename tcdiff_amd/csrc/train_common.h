// Shared pieces of the training-side kernels (train_ops.hip, attention_train.hip), gfx950.
//
// Dropout.  The reference draws its dropout masks from torch's generator (nn.Dropout / F.dropout at
// model/model.py:98,103,240,244-245,383,396,400-401 and inside nn.MultiheadAttention, :190-192).  A backward pass
// has to see the SAME mask as its forward, so the masks here are a pure function of (seed, site, element):
//     key  = fmix32(seed0 ^ (0x9E3779B9 * (site + 1))) ^ seed1
//     keep = fmix32((x * 0x9E3779B1) ^ key) >= thr,           thr = floor(p * 2^32),  kept values times 1 / (1 - p)
// with fmix32 = MurmurHash3's 32-bit finaliser and x = the element's flat index in the reference's tensor at that
// site (32-bit wrap-around).  Nothing is stored: forward and backward kernels regenerate the bits, and the CPU oracle
// (oracle/tcdiff_oracle.py dropout_mask) evaluates the same function in numpy, which is how the parity tests feed the
// real reference identical masks.  Sites (tcdiff_hip.h TC_SITE_*): encoder layer i: 4 i + {0 attention weights,
// 1 dropout1, 2 feed-forward inner, 3 dropout2}; decoder layer l: 16 + 8 l + {0 self-attention weights, 1 self fc out,
// 2 dropout1, 3 cross-attention weights, 4 cross fc out, 5 dropout2, 6 feed-forward inner, 7 dropout3}.
#pragma once
#include "common.h"

DEVINL uint32_t tc_fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

struct DropCtx {
    uint32_t key, thr;
    float scale;
};
// seed: DEVICE int[2] (so that a captured graph can be replayed with new masks) or NULL (= {0, 0})
DEVINL DropCtx drop_ctx(const int* seed, int site, uint32_t thr, float scale) {
    DropCtx d;
    const uint32_t s0 = seed ? (uint32_t)seed[0] : 0u, s1 = seed ? (uint32_t)seed[1] : 0u;
    d.key = tc_fmix32(s0 ^ (0x9E3779B9u * (uint32_t)(site + 1))) ^ s1;
    d.thr = thr;
    d.scale = scale;
    return d;
}
DEVINL bool drop_keep(const DropCtx& d, uint32_t x) { return tc_fmix32((x * 0x9E3779B1u) ^ d.key) >= d.thr; }
DEVINL float drop_apply(const DropCtx& d, uint32_t x, float v) { return drop_keep(d, x) ? v * d.scale : 0.0f; }

// ---- activation derivatives (the forward forms are common.h's) ------------------------------------------------------
// gelu'(x) = Phi(x) + x phi(x)   (F.gelu, exact erf form: TCDiff.py:85).  Phi from the erf of common.h::gelu_erf
// (Abramowitz-Stegun 7.1.28, |error| <= 3e-7: erf(|z|) = 1 - r, r = 1 / poly(|z|)^16), phi by one v_exp_f32 -- libm's erff +
// expf were ~100 instructions per element in the 14.7 M-element backward of every feed-forward block.
DEVINL float gelu_grad(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    float p = fmaf(0.0000430638f, z, 0.0002765672f);
    p = fmaf(p, z, 0.0001520143f);
    p = fmaf(p, z, 0.0092705272f);
    p = fmaf(p, z, 0.0422820123f);
    p = fmaf(p, z, 0.0705230784f);
    p = fmaf(p, z, 1.0f);
    p = p * p; p = p * p; p = p * p; p = p * p;
    const float hr = 0.5f * __builtin_amdgcn_rcpf(p);
    const float cdf = x >= 0.0f ? 1.0f - hr : hr;
    const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
    return fmaf(x, pdf, cdf);
}
// mish(x) = x tanh(softplus(x))  (nn.Mish; softplus threshold 20 as torch)
DEVINL float mish_grad(float x) {
    const float sp = softplus_t(x);
    const float th = tanhf(sp);
    const float sg = 1.0f / (1.0f + expf(-x));          // d softplus / dx (= 1 beyond the threshold to fp32 precision)
    return th + x * (1.0f - th * th) * sg;
}
DEVINL float silu_grad(float x) {
    const float sg = 1.0f / (1.0f + expf(-x));
    return sg * (1.0f + x * (1.0f - sg));
}
DEVINL float act_grad(float x, int act) {
    switch (act) {
        case ACT_RELU: return x > 0.0f ? 1.0f : 0.0f;
        case ACT_GELU: return gelu_grad(x);
        case ACT_MISH: return mish_grad(x);
        case ACT_SILU: return silu_grad(x);
        default: return 1.0f;
    }
}
