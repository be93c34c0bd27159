"""CPU oracle for the TCDiff denoising hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch functional restatement (torch CPU, fp32) of the
reference algorithm for the path named by BASELINE.json: the ``DanceDecoder``
denoiser and the ``GaussianDiffusion`` samplers.  Nothing here is shipped or
measured as the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it (as the checker / the
reported CPU baseline).  The product path (``tcdiff_amd``) never imports it.

Parity status: PINNED.  ``tests/golden/make_golden*.py`` import the real reference
from /root/reference (in the build container only, through ``oracle/refload.py``)
and store its outputs as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
holds every function below to those vectors on any box.

All citations are file:line in the reference repo (Da1yuqin/TCDiff @ 2025-10-17).
Weights are passed as a flat ``state_dict`` (name -> tensor) with the reference's
own key names, so a reference checkpoint feeds the oracle unchanged.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]

DK = 64  # SBI_MSA hard-codes d_k = 64 (model/model.py:55)


# ----------------------------------------------------------------------------------------------
# optional bf16 OPERAND-ROUNDING emulation (test infrastructure for the bf16 training step, round 5)
# ----------------------------------------------------------------------------------------------
# The product's bf16 mode rounds every GEMM / attention OPERAND to bf16 (fp32 accumulation), keeps the gradients of T-typed
# activations in bf16 and -- in the training engine -- stores the GEMM outputs it keeps for the backward (z1 .. z4, Q / K / V, O) in
# bf16.  `operand_rounding(...)` re-routes this module's own F.linear / torch.matmul calls through
# versions that round at exactly those sites (values stay fp32 tensors on the bf16 grid), so that autograd through the oracle
# yields "the fp32 algorithm with the kernels' rounding points" -- the thing a bf16 step can be held to, as opposed to the exact
# fp32 gradient it cannot reach.  Not a reference restatement: the reference has no reduced-precision path.
import contextlib
import types


def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


class _RoundF(torch.autograd.Function):      # value rounded in the forward, gradient passed through
    @staticmethod
    def forward(ctx, x):
        return _bf(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundB(torch.autograd.Function):      # identity in the forward, gradient rounded
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _bf(g)


@contextlib.contextmanager
def operand_rounding(fwd: bool = True, bwd: bool = True, out: bool = False):
    """fwd: GEMM / attention operands rounded in the forward; bwd: gradients of linear inputs / outputs and of the attention
    products' operands rounded (dX, dY, dQ, dK, dV, dP, dO; parameter gradients stay fp32); out: GEMM outputs rounded too."""
    g = globals()
    real_F, real_torch = g["F"], g["torch"]
    rf = (lambda x: _RoundF.apply(x)) if fwd else (lambda x: x)
    rb = (lambda x: _RoundB.apply(x)) if bwd else (lambda x: x)

    def linear(x, w, b=None):
        y = rb(real_F.linear(rf(rb(x)), rf(w), b))
        return _RoundF.apply(y) if out else y

    def matmul(a, b):
        return rb(real_torch.matmul(rf(rb(a)), rf(rb(b))))
    Fp = types.SimpleNamespace(**{k: getattr(real_F, k) for k in dir(real_F) if not k.startswith("__")})
    Fp.linear = linear
    Tp = types.SimpleNamespace(**{k: getattr(real_torch, k) for k in dir(real_torch) if not k.startswith("__")})
    Tp.matmul = matmul
    g["F"], g["torch"] = Fp, Tp
    try:
        yield
    finally:
        g["F"], g["torch"] = real_F, real_torch


# ----------------------------------------------------------------------------------------------
# schedule tables  (model/utils.py:67-99, model/diffusion.py:109-169)
# ----------------------------------------------------------------------------------------------
def cosine_betas(n_timestep: int, s: float = 8e-3) -> np.ndarray:
    """model/utils.py:78-86 -- fp64 cosine schedule, clipped to [0, 0.999]."""
    ts = torch.arange(n_timestep + 1, dtype=torch.float64) / n_timestep + s
    ac = torch.cos(ts / (1 + s) * np.pi / 2).pow(2)
    ac = ac / ac[0]
    betas = 1 - ac[1:] / ac[:-1]
    return np.clip(betas.numpy(), 0, 0.999)


def linear_betas(n_timestep: int, start: float = 1e-4, end: float = 2e-2) -> np.ndarray:
    """model/utils.py:70-76."""
    return (torch.linspace(start ** 0.5, end ** 0.5, n_timestep, dtype=torch.float64) ** 2).numpy()


def make_tables(n_timestep: int = 1000, schedule: str = "cosine") -> Dict[str, torch.Tensor]:
    """The 13 fp32 [T] buffers of GaussianDiffusion.__init__ (model/diffusion.py:109-169).

    betas are cast to fp32 FIRST (``torch.Tensor(np_f64)``), every derived table is then
    computed in fp32 (``np.sqrt`` on a torch tensor dispatches to torch.sqrt)."""
    b64 = cosine_betas(n_timestep) if schedule == "cosine" else linear_betas(n_timestep)
    betas = torch.tensor(b64, dtype=torch.float64).to(torch.float32)
    alphas = 1.0 - betas
    ac = torch.cumprod(alphas, dim=0)
    ac_prev = torch.cat([torch.ones(1), ac[:-1]])
    post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
    t = {
        "betas": betas,
        "alphas_cumprod": ac,
        "alphas_cumprod_prev": ac_prev,
        "sqrt_alphas_cumprod": torch.sqrt(ac),
        "sqrt_one_minus_alphas_cumprod": torch.sqrt(1.0 - ac),
        "log_one_minus_alphas_cumprod": torch.log(1.0 - ac),
        "sqrt_recip_alphas_cumprod": torch.sqrt(1.0 / ac),
        "sqrt_recipm1_alphas_cumprod": torch.sqrt(1.0 / ac - 1),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": torch.log(torch.clamp(post_var, min=1e-20)),
        # the reference applies numpy's sqrt to torch tensors here (model/diffusion.py:155,159); numpy's
        # fp32 sqrt and torch's differ in the last bit for some inputs, so the same call is used
        "posterior_mean_coef1": betas * torch.from_numpy(np.sqrt(ac_prev.numpy())) / (1.0 - ac),
        "posterior_mean_coef2": (1.0 - ac_prev) * torch.from_numpy(np.sqrt(alphas.numpy())) / (1.0 - ac),
        "p2_loss_weight": (1 + ac / (1 - ac)) ** -0.0,
    }
    return t


# ----------------------------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------------------------
def abs_pos(x: torch.Tensor, sd, drop=None, site: int = 8) -> torch.Tensor:
    """DanceDecoder.abs_pos_encoding (model/model.py:441-448,564,580): identity with rotary embeddings; with use_rotary=False
    PositionalEncoding (model/utils.py:11-32, batch_first): dropout(x + pe[:len]) -- the module's own nn.Dropout (same p as the
    model's, model/model.py:446-448), sites 8 (motion tokens, :564) and 9 (music tokens, :580) of the DropPlan."""
    pe = sd.get("abs_pos_encoding.pe")
    if pe is None:
        return x
    y = x + pe[: x.shape[-2], 0, :].to(x.dtype)
    return y if drop is None else drop(y, site)


def rotary(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """RotaryEmbedding.rotate_queries_or_keys (model/rotary_embedding_torch.py:107-113,46-59,39-43).

    Position = index along dim -2; pair j=(2j,2j+1) rotates by angle pos*freqs[j]:
    (y0, y1) = (x0 cos - x1 sin, x1 cos + x0 sin)."""
    if freqs is None:       # use_rotary=False (model/model.py:231,375,387-388): no rotation
        return x
    n = x.shape[-2]
    ang = torch.arange(n, dtype=freqs.dtype)[:, None] * freqs[None, :]  # (n, D/2)
    ang = ang.repeat_interleave(2, dim=-1)  # (n, D): [a0,a0,a1,a1,...]
    xp = x.reshape(*x.shape[:-1], -1, 2)
    rh = torch.stack((-xp[..., 1], xp[..., 0]), dim=-1).reshape(x.shape)
    return x * ang.cos() + rh * ang.sin()


def sinusoidal_emb(times: torch.Tensor, dim: int) -> torch.Tensor:
    """SinusoidalPosEmb (model/utils.py:36-48)."""
    half = dim // 2
    f = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    e = times[:, None] * f[None, :]
    return torch.cat((e.sin(), e.cos()), dim=-1)


def mish(x: torch.Tensor) -> torch.Tensor:
    return x * torch.tanh(F.softplus(x))


# ----------------------------------------------------------------------------------------------
# train-mode dropout: the product's counter hash (tcdiff_amd/csrc/train_common.h), evaluated in numpy
# ----------------------------------------------------------------------------------------------
def _fmix32(h: np.ndarray) -> np.ndarray:
    """MurmurHash3's 32-bit finaliser on uint32 arrays (wrap-around arithmetic through uint64)."""
    h = h.astype(np.uint64)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    return h


def dropout_keep(seed, site: int, shape, p: float) -> torch.Tensor:
    """keep mask (bool, `shape`) of dropout site `site`: element with C-order flat index x is kept iff
    fmix32((x * 0x9E3779B1) ^ key) >= floor(p * 2^32), key = fmix32(seed0 ^ (0x9E3779B9 * (site + 1))) ^ seed1."""
    n = int(np.prod(shape))
    s0, s1 = np.uint64(seed[0] & 0xFFFFFFFF), np.uint64(seed[1] & 0xFFFFFFFF)
    key = _fmix32(np.array([s0 ^ np.uint64((0x9E3779B9 * (site + 1)) & 0xFFFFFFFF)], dtype=np.uint64))[0] ^ s1
    x = np.arange(n, dtype=np.uint64) & np.uint64(0xFFFFFFFF)
    h = _fmix32(((x * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)) ^ key)
    return torch.from_numpy((h >= np.uint64(int(p * 4294967296.0))).reshape(shape))


class DropPlan:
    """Train-mode dropout with the product's masks: plan(x, site) = x * keep / (1 - p)  (nn.Dropout, F.dropout).
    Sites: PositionalEncoding (use_rotary=False) 8 motion tokens, 9 music tokens (model/utils.py:32 at model/model.py:564,580);
    encoder layer i -> 4 i + {0 attention weights, 1 dropout1, 2 feed-forward inner, 3 dropout2}; decoder layer l
    -> 16 + 8 l + {0 self weights, 1 self fc out, 2 dropout1, 3 cross weights, 4 cross fc out, 5 dropout2, 6 inner,
    7 dropout3}  (model/model.py:98,103,240,244-245,383,396,400-401)."""

    def __init__(self, seed, p: float):
        self.seed, self.p = (int(seed[0]), int(seed[1])), float(p)

    def __call__(self, x: torch.Tensor, site: int) -> torch.Tensor:
        if self.p <= 0.0:
            return x
        keep = dropout_keep(self.seed, site, tuple(x.shape), self.p)
        return torch.where(keep, x * np.float32(1.0 / (1.0 - self.p)), torch.zeros_like(x))


def _nodrop(x, site):
    return x


def layer_norm(x, sd: SD, prefix: str, eps: float) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def linear(x, sd: SD, prefix: str, bias: bool = True) -> torch.Tensor:
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"] if bias else None)


def _heads(x: torch.Tensor, n_head: int) -> torch.Tensor:
    b, n, _ = x.shape
    return x.view(b, n, n_head, -1).transpose(1, 2)  # (b,h,n,dk)


KERNEL_DELTA = False     # (experiment, eval mode only) the attention core with the backward the HIP kernels run: see _AttnKernelDelta


class _AttnKernelDelta(torch.autograd.Function):
    """softmax(q k^T) v with bf16 operands and the FLASH-style backward of csrc/attention_train.hip: the softmax Jacobian's row term
    delta_i = sum_j P_ij dP_ij is taken as rowsum(dO * O) from the STORED (bf16) output -- equal in exact arithmetic, but O was
    rounded to 8 bits and the term it enters, P (dP - delta), is a cancellation.  Autograd through torch.softmax (the plain
    operand_rounding emulation) computes delta from the fp32 P and dP.  tests/golden/make_golden_bf16_draws.py --kernel-delta asks
    whether this is what separates the kernels' w_qs / w_ks gradients from the emulated draws at the config-1 shape."""
    @staticmethod
    def forward(ctx, q, k, v):
        qb, kb, vb = _bf(q), _bf(k), _bf(v)
        p = torch.softmax(torch.matmul(qb, kb.transpose(2, 3)), dim=-1)
        o = torch.matmul(_bf(p), vb)
        ctx.save_for_backward(qb, kb, vb, p, _bf(o))
        return o

    @staticmethod
    def backward(ctx, d_o):
        qb, kb, vb, p, ob = ctx.saved_tensors
        dob = _bf(d_o)
        dv = torch.matmul(_bf(p).transpose(2, 3), dob)
        dp = torch.matmul(dob, vb.transpose(2, 3))
        delta = (dob * ob).sum(-1, keepdim=True)
        ds = _bf(p * (dp - delta))
        return _bf(torch.matmul(ds, kb)), _bf(torch.matmul(ds.transpose(2, 3), qb)), _bf(dv)


def sbi_msa(q_in, k_in, v_in, sd: SD, prefix: str, n_head: int, drop=_nodrop, site0: int = 0) -> torch.Tensor:
    """SBI_MSA.forward with trj_dist=None (model/model.py:71-107); drop: train-mode dropout on the softmax weights (site0)
    and on the fc output (site0 + 1), :98,103.

    softmax((Q/8) K^T) V, bias-free projections, bias-free fc, LayerNorm(eps=1e-6).
    The ``indexed_matrix`` product (:82-83) does not reach the output and is omitted."""
    q = _heads(F.linear(q_in, sd[prefix + ".w_qs.weight"]), n_head)
    k = _heads(F.linear(k_in, sd[prefix + ".w_ks.weight"]), n_head)
    v = _heads(F.linear(v_in, sd[prefix + ".w_vs.weight"]), n_head)
    if KERNEL_DELTA and (drop is _nodrop or getattr(drop, "p", 1.0) <= 0.0):
        o = _AttnKernelDelta.apply(q / (DK ** 0.5), k, v).transpose(1, 2).reshape(q_in.shape[0], q_in.shape[1], -1)
    else:
        att = drop(torch.softmax(torch.matmul(q / (DK ** 0.5), k.transpose(2, 3)), dim=-1), site0)
        o = torch.matmul(att, v).transpose(1, 2).reshape(q_in.shape[0], q_in.shape[1], -1)
    o = drop(F.linear(o, sd[prefix + ".fc.weight"]), site0 + 1)
    return layer_norm(o, sd, prefix + ".layer_norm", 1e-6)


def film(t: torch.Tensor, sd: SD, prefix: str):
    """DenseFiLM (model/model.py:154-168): Linear(Mish(t)) -> (scale, shift), each (B,1,D)."""
    p = linear(mish(t), sd, prefix + ".block.1")[:, None, :]
    return p.chunk(2, dim=-1)


def affine(x, scale_shift):
    """featurewise_affine (model/model.py:171-173)."""
    scale, shift = scale_shift
    return (scale + 1) * x + shift


FF_ACTIVATION = None     # tests of the constructor's `activation` option set this (a callable); None = F.gelu, the production value


def gelu(x):
    # the feed-forward activation of the encoder / decoder layers (model/model.py:244,400); exact-erf GELU: TCDiff.py:85 passes F.gelu
    return F.gelu(x) if FF_ACTIVATION is None else FF_ACTIVATION(x)


def encoder_layer(x, sd: SD, prefix: str, freqs, n_head: int, drop=_nodrop, site0: int = 0) -> torch.Tensor:
    """TransformerEncoderLayer, norm_first, eval (model/model.py:211-245).

    nn.MultiheadAttention with packed in_proj (+bias), q=k=rot(LN1 x), v=LN1 x, out_proj with bias."""
    d = x.shape[-1]
    h = layer_norm(x, sd, prefix + ".norm1", 1e-5)
    qk = rotary(h, freqs)
    w, b = sd[prefix + ".self_attn.in_proj_weight"], sd[prefix + ".self_attn.in_proj_bias"]
    q = _heads(F.linear(qk, w[:d], b[:d]), n_head)
    k = _heads(F.linear(qk, w[d:2 * d], b[d:2 * d]), n_head)
    v = _heads(F.linear(h, w[2 * d:], b[2 * d:]), n_head)
    att = drop(torch.softmax(torch.matmul(q, k.transpose(2, 3)) / math.sqrt(d // n_head), dim=-1), site0)
    o = torch.matmul(att, v).transpose(1, 2).reshape(x.shape)
    x = x + drop(linear(o, sd, prefix + ".self_attn.out_proj"), site0 + 1)
    h = layer_norm(x, sd, prefix + ".norm2", 1e-5)
    return x + drop(linear(drop(gelu(linear(h, sd, prefix + ".linear1")), site0 + 2), sd, prefix + ".linear2"), site0 + 3)


def decoder_layer(x, mem, t, sd: SD, prefix: str, freqs, n_head: int, drop=_nodrop, site0: int = 0) -> torch.Tensor:
    """FiLMTransformerDecoderLayer.forward, norm_first, eval (model/model.py:323-344,371).

    The traj_Modulation result (:346-355) is discarded by ``return x`` (:371): not computed."""
    h = layer_norm(x, sd, prefix + ".norm1", 1e-5)
    qk = rotary(h, freqs)
    a = drop(sbi_msa(qk, qk, h, sd, prefix + ".self_attn", n_head, drop, site0), site0 + 2)
    x = x + affine(a, film(t, sd, prefix + ".film1"))
    h = layer_norm(x, sd, prefix + ".norm2", 1e-5)
    c = sbi_msa(rotary(h, freqs), rotary(mem, freqs), mem, sd, prefix + ".multihead_attn", n_head, drop, site0 + 3)
    x = x + affine(drop(c, site0 + 5), film(t, sd, prefix + ".film2"))
    h = layer_norm(x, sd, prefix + ".norm3", 1e-5)
    f = drop(linear(drop(gelu(linear(h, sd, prefix + ".linear1")), site0 + 6), sd, prefix + ".linear2"), site0 + 7)
    x = x + affine(f, film(t, sd, prefix + ".film3"))
    return linear(layer_norm(x, sd, prefix + ".norm4", 1e-5), sd, prefix + ".linear3")  # NO residual (:344)


# ----------------------------------------------------------------------------------------------
# DanceDecoder
# ----------------------------------------------------------------------------------------------
def infer_config(sd: SD) -> dict:
    latent = sd["input_projection.weight"].shape[0]
    nfeats = sd["input_projection.weight"].shape[1]
    dn = sd["relative_projection_layer.0.weight"].shape[1] // latent
    seq_len = sd["null_cond_embed"].shape[1]
    n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("seqTransDecoder.stack."))
    n_head = sd["seqTransDecoder.stack.0.self_attn.w_qs.weight"].shape[0] // DK
    return dict(latent=latent, nfeats=nfeats, dn=dn, seq_len=seq_len, n_layers=n_layers, n_head=n_head)


def music_branch(sd: SD, cond_embed: torch.Tensor, cfg: dict, drop=_nodrop):
    """Step-invariant part of DanceDecoder.forward (model/model.py:572-583,593-597):
    returns (cond_tokens (B,S,D) before the keep-mask select, cond_hidden (B,D) for kept clips)."""
    b, clen, _ = cond_embed.shape
    if clen % 2 == 1:
        cond_embed = cond_embed[:, :-1, :]
    c = cond_embed.reshape(b, clen // 2, -1).float()
    tok = linear(F.relu(linear(c, sd, "cond_projection.0")), sd, "cond_projection.2")
    tok = abs_pos(tok, sd, drop, 9)        # :580
    for i in range(2):
        tok = encoder_layer(tok, sd, f"cond_encoder.{i}", sd.get("rotary.freqs"), cfg["n_head"], drop, 4 * i)
    return tok


def cond_hidden_of(sd: SD, tokens: torch.Tensor) -> torch.Tensor:
    """non_attn_cond_projection on mean-pooled tokens (model/model.py:593-597,496-501)."""
    p = tokens.mean(dim=-2)
    p = layer_norm(p, sd, "non_attn_cond_projection.0", 1e-5)
    return linear(F.silu(linear(p, sd, "non_attn_cond_projection.1")), sd, "non_attn_cond_projection.3")


def decoder_forward(sd: SD, x, cond_embed, times, cond_drop_prob: float = 0.0,
                    keep_mask: Optional[torch.Tensor] = None, drop=_nodrop) -> torch.Tensor:
    """DanceDecoder.forward, trj_dist=None (model/model.py:548-624); eval mode, or train mode with `drop` a DropPlan.

    keep_mask: optional explicit bool (B,) (else prob_mask_like semantics, model/utils.py:52-58)."""
    cfg = infer_config(sd)
    D, S, dn, nf = cfg["latent"], cfg["seq_len"], cfg["dn"], cfg["nfeats"]
    B = x.shape[0]
    x = x.reshape(B, -1, nf)
    # :560-561 input projection + fusion projection over per-frame concatenated dancers
    x = linear(x, sd, "input_projection")
    f = x.reshape(B, S, D * dn)
    f = F.relu(linear(f, sd, "relative_projection_layer.0"))
    f = F.relu(linear(f, sd, "relative_projection_layer.2"))
    x = linear(f, sd, "relative_projection_layer.4").reshape(B, dn * S, D)
    x = abs_pos(x, sd, drop, 8)            # :564
    # :567-569 keep mask
    if keep_mask is None:
        p = 1 - cond_drop_prob
        if p == 1:
            keep_mask = torch.ones(B, dtype=torch.bool)
        elif p == 0:
            keep_mask = torch.zeros(B, dtype=torch.bool)
        else:
            keep_mask = torch.zeros(B).float().uniform_(0, 1) < p
    # :572-589 music tokens, null select
    tok = music_branch(sd, cond_embed, cfg, drop)
    tok = torch.where(keep_mask[:, None, None], tok, sd["null_cond_embed"].to(tok.dtype))
    # :593-610 pooled hidden, null select
    ch = cond_hidden_of(sd, tok)
    ch = torch.where(keep_mask[:, None], ch, sd["null_cond_hidden"].to(ch.dtype))
    # :601-612 time path
    th = mish(linear(sinusoidal_emb(times, D), sd, "time_mlp.1"))
    t = linear(th, sd, "to_time_cond.0") + ch
    ttok = linear(th, sd, "to_time_tokens.0").reshape(B, 2, D)
    # :615-616 memory
    mem = layer_norm(torch.cat((tok, ttok), dim=-2), sd, "norm_cond", 1e-5)
    # :621 decoder stack, :623 final layer
    for i in range(cfg["n_layers"]):
        x = decoder_layer(x, mem, t, sd, f"seqTransDecoder.stack.{i}", sd.get("rotary.freqs"), cfg["n_head"], drop, 16 + 8 * i)
    return linear(x, sd, "final_layer")


def guided_forward(sd: SD, x, cond_embed, times, guidance_weight) -> torch.Tensor:
    """DanceDecoder.guided_forward (model/model.py:542-546)."""
    unc = decoder_forward(sd, x, cond_embed, times, cond_drop_prob=1)
    con = decoder_forward(sd, x, cond_embed, times, cond_drop_prob=0)
    return unc + (con - unc) * guidance_weight


# ----------------------------------------------------------------------------------------------
# GaussianDiffusion samplers
# ----------------------------------------------------------------------------------------------
NoiseFn = Callable[[int, torch.Size], torch.Tensor]  # (call index, shape) -> N(0,1) tensor


def _default_noise(_i, shape):
    return torch.randn(shape)


def ddpm_guidance_weight(i: int, n_timestep: int, w: float) -> float:
    """Guidance clipping of p_mean_variance (model/diffusion.py:219-224)."""
    if i > 1.0 * n_timestep:
        return min(w, 0)
    if i < 0.1 * n_timestep:
        return min(w, 1)
    return w


def p_sample(sd: SD, tab, x, cond, i: int, n_timestep: int, w: float, eps: torch.Tensor):
    """p_sample + p_mean_variance + q_posterior, predict_epsilon=False, clip_denoised=True
    (model/diffusion.py:206-252)."""
    B = x.shape[0]
    t = torch.full((B,), i, dtype=torch.long)
    x0 = guided_forward(sd, x, cond, t, ddpm_guidance_weight(i, n_timestep, w)).clamp(-1.0, 1.0)
    mean = tab["posterior_mean_coef1"][i] * x0 + tab["posterior_mean_coef2"][i] * x
    logvar = tab["posterior_log_variance_clipped"][i]
    nonzero = 0.0 if i == 0 else 1.0
    return mean + nonzero * (0.5 * logvar).exp() * eps, x0


def p_sample_loop(sd: SD, shape, cond, noise: Optional[torch.Tensor] = None, n_timestep: int = 1000,
                  guidance_weight: float = 2, schedule: str = "cosine", start_point: Optional[int] = None,
                  step_noise: NoiseFn = _default_noise, return_diffusion: bool = False):
    """GaussianDiffusion.p_sample_loop (model/diffusion.py:255-286).

    ``step_noise(i, shape)`` supplies the randn_like draw of step i (drawn at i == 0 too, :246)."""
    tab = make_tables(n_timestep, schedule)
    start = n_timestep if start_point is None else start_point
    x = torch.randn(shape) if noise is None else noise.clone()
    diff = [x]
    for i in reversed(range(0, start)):
        x, _ = p_sample(sd, tab, x, cond, i, n_timestep, guidance_weight, step_noise(i, x.shape))
        if return_diffusion:
            diff.append(x)
    return (x, diff) if return_diffusion else x


def ddim_time_pairs(n_timestep: int, sampling_timesteps: int = 50):
    """model/diffusion.py:389-391: linspace(-1, T-1, 51).int() reversed, consecutive pairs."""
    times = torch.linspace(-1, n_timestep - 1, steps=sampling_timesteps + 1)
    times = list(reversed(times.int().tolist()))
    return list(zip(times[:-1], times[1:]))


def model_predictions(sd: SD, tab, x, cond, time: int, w: float):
    """model/diffusion.py:195-204 with clip_x_start=True."""
    t = torch.full((x.shape[0],), time, dtype=torch.long)
    x0 = guided_forward(sd, x, cond, t, w).clamp(-1.0, 1.0)
    pred_noise = (tab["sqrt_recip_alphas_cumprod"][time] * x - x0) / tab["sqrt_recipm1_alphas_cumprod"][time]
    return pred_noise, x0


def _overwrite_traj(x, x0_traj, frames: int, nf: int):
    """x[:,:,:,[4,5]] = x_0[:,:,:,[0,1]] on the (b, frames, dn, nf) view (model/diffusion.py:396-403)."""
    b, seq, _ = x.shape
    xv = x.reshape(b, frames, seq // frames, nf)
    xv[:, :, :, [4, 5]] = x0_traj.reshape(b, frames, seq // frames, -1)[:, :, :, [0, 1]]
    return xv.reshape(b, seq, nf)


def ddim_sample(sd: SD, shape, cond, x_0: Optional[torch.Tensor] = None, n_timestep: int = 1000,
                guidance_weight: float = 2, schedule: str = "cosine", init_noise: Optional[torch.Tensor] = None,
                step_noise: NoiseFn = _default_noise, frames: int = 150):
    """GaussianDiffusion.ddim_sample, 50 steps, eta=1 (model/diffusion.py:386-442).

    ``frames`` is the reference's hard-coded 150 (:399-400)."""
    tab = make_tables(n_timestep, schedule)
    ac = tab["alphas_cumprod"]
    nf = shape[-1]
    x = torch.randn(shape) if init_noise is None else init_noise.clone()
    if x_0 is not None:
        x = _overwrite_traj(x, x_0, frames, nf)
    for time, time_next in ddim_time_pairs(n_timestep):
        pred_noise, x_start = model_predictions(sd, tab, x, cond, time, guidance_weight)
        if time_next < 0:
            x = x_start
            continue
        alpha, alpha_next = ac[time], ac[time_next]
        sigma = 1 * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
        c = (1 - alpha_next - sigma ** 2).sqrt()
        x = x_start * alpha_next.sqrt() + c * pred_noise + sigma * step_noise(time, x.shape)
        if x_0 is not None:
            x = _overwrite_traj(x, x_0, frames, nf)
    if x_0 is not None:
        x = _overwrite_traj(x, x_0, frames, nf)
    return x


def long_inpaint_loop(sd: SD, shape, cond, noise: Optional[torch.Tensor] = None, n_timestep: int = 1000,
                      guidance_weight: float = 2, schedule: str = "cosine", start_point: Optional[int] = None,
                      step_noise: NoiseFn = _default_noise):
    """GaussianDiffusion.long_inpaint_loop (model/diffusion.py:560-608): DDPM steps with the first half of every
    clip's tokens overwritten by the second half of the previous clip after each step but the last."""
    if shape[0] == 1:
        return p_sample_loop(sd, shape, cond, noise, n_timestep, guidance_weight, schedule, start_point, step_noise)
    tab = make_tables(n_timestep, schedule)
    start = n_timestep if start_point is None else start_point
    x = torch.randn(shape) if noise is None else noise.clone()
    half = x.shape[1] // 2
    for i in reversed(range(0, start)):
        x, _ = p_sample(sd, tab, x, cond, i, n_timestep, guidance_weight, step_noise(i, x.shape))
        if i > 0:
            x[1:, :half] = x[:-1, half:].clone()
    return x


FOOT_JOINTS = (1, 2, 3, 4, 5, 7, 8, 10, 11)   # model/diffusion.py:307,338,375


def _overwrite_footwork(x, x0, frames: int, nf: int):
    """Trajectory channels [4,5] <- x_0[...,[0,1]] and the lower-body 6-D rotations of frames 75:120 <- x_0
    (model/diffusion.py:300-309, 335-341) on the (b, frames, dn, nf) view."""
    b, seq, _ = x.shape
    xv = x.reshape(b, frames, seq // frames, nf)
    x0v = x0.reshape(b, frames, seq // frames, nf)
    xv[:, :, :, [4, 5]] = x0v[:, :, :, [0, 1]]
    for i in FOOT_JOINTS:
        sl = slice(4 + 3 + (i - 1) * 6, 4 + 3 + i * 6)
        xv[:, 75:120, :, sl] = x0v[:, 75:120, :, sl]
    return xv.reshape(b, seq, nf)


def ddim_sample_footwork(sd: SD, shape, cond, x_0: Optional[torch.Tensor] = None, n_timestep: int = 1000,
                         guidance_weight: float = 2, schedule: str = "cosine",
                         init_noise: Optional[torch.Tensor] = None, step_noise: NoiseFn = _default_noise,
                         frames: int = 150):
    """GaussianDiffusion.ddim_sample_Footwork (model/diffusion.py:289-383): DDIM (50 steps, eta 1) with the trajectory
    channels and the lower-body rotations of frames 75:120 re-imposed after every step, and a 10-frame linear blend at
    the segment borders at the end.  x_0 is a full motion tensor (b, seq*dn, nf)."""
    tab = make_tables(n_timestep, schedule)
    ac = tab["alphas_cumprod"]
    nf = shape[-1]
    x = torch.randn(shape) if init_noise is None else init_noise.clone()
    if x_0 is not None:
        x = _overwrite_footwork(x, x_0, frames, nf)
    for time, time_next in ddim_time_pairs(n_timestep):
        pred_noise, x_start = model_predictions(sd, tab, x, cond, time, guidance_weight)
        if time_next < 0:
            x = x_start
            continue
        alpha, alpha_next = ac[time], ac[time_next]
        sigma = 1 * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
        c = (1 - alpha_next - sigma ** 2).sqrt()
        x = x_start * alpha_next.sqrt() + c * pred_noise + sigma * step_noise(time, x.shape)
        if x_0 is not None:
            x = _overwrite_footwork(x, x_0, frames, nf)
    if x_0 is not None:
        b, seq, _ = x.shape
        xv = x.reshape(b, frames, seq // frames, nf)
        x0v = x_0.reshape(b, frames, seq // frames, nf)
        xv[:, :, :, [4, 5]] = x0v[:, :, :, [0, 1]]
        width = 10
        wgt = torch.from_numpy(np.linspace(0, 1, width)).to(xv)[None, :, None, None]      # :364-365
        for i in FOOT_JOINTS:
            sl = slice(4 + 3 + (i - 1) * 6, 4 + 3 + i * 6)
            xv[:, 75:75 + width, :, sl] = wgt * x0v[:, 75:75 + width, :, sl] + (1 - wgt) * xv[:, 75:75 + width, :, sl]
            xv[:, 75 + width:-width, :, sl] = x0v[:, 75 + width:-width, :, sl]              # :373: up to frame -10, not 110
            xv[:, 120 - width:120, :, sl] = (1 - wgt) * x0v[:, 120 - width:120, :, sl] + wgt * xv[:, 120 - width:120, :, sl]
        x = xv.reshape(b, seq, nf)
    return x


def inpaint_loop(sd: SD, shape, cond, noise: Optional[torch.Tensor], mask: torch.Tensor, value: torch.Tensor,
                 n_timestep: int = 1000, guidance_weight: float = 2, schedule: str = "cosine",
                 start_point: Optional[int] = None, step_noise: NoiseFn = _default_noise,
                 q_noise: NoiseFn = _default_noise):
    """GaussianDiffusion.inpaint_loop (model/diffusion.py:519-557): after every p_sample the constrained entries are
    replaced by q_sample(value, i-1) (by x itself at i == 0).  ``q_noise(i, shape)`` is the randn_like drawn inside
    q_sample in step i (not drawn at i == 0)."""
    tab = make_tables(n_timestep, schedule)
    start = n_timestep if start_point is None else start_point
    x = torch.randn(shape) if noise is None else noise.clone()
    for i in reversed(range(0, start)):
        x, _ = p_sample(sd, tab, x, cond, i, n_timestep, guidance_weight, step_noise(i, x.shape))
        if i > 0:
            t = torch.full((x.shape[0],), i - 1, dtype=torch.long)
            value_ = q_sample(tab, value, t, q_noise(i, x.shape))
        else:
            value_ = x
        x = value_ * mask + (1.0 - mask) * x
    return x


def long_ddim_sample(sd: SD, shape, cond, x_0, seq_len: int, n_timestep: int = 1000,
                     guidance_weight: float = 2, schedule: str = "cosine",
                     init_noise: Optional[torch.Tensor] = None, step_noise: NoiseFn = _default_noise):
    """GaussianDiffusion.long_ddim_sample (model/diffusion.py:446-515): weight ramp (:454) and
    window coupling x[1:, :half] = x[:-1, half:] on the (b, seq_len, dn, nf) view (:502-506).
    x_0 is (b, seq, dn, 3) here (:461-462)."""
    if shape[0] == 1:
        return ddim_sample(sd, shape, cond, None, n_timestep, guidance_weight, schedule, init_noise, step_noise)
    tab = make_tables(n_timestep, schedule)
    ac = tab["alphas_cumprod"]
    nf = shape[-1]
    weights = np.clip(np.linspace(0, guidance_weight * 2, 50), None, guidance_weight)
    x = torch.randn(shape) if init_noise is None else init_noise.clone()
    if x_0 is not None:
        x = _overwrite_traj(x, x_0.reshape(shape[0], -1, 3), x_0.shape[1], nf)
    half = seq_len // 2
    for (time, time_next), w in zip(ddim_time_pairs(n_timestep), weights):
        pred_noise, x_start = model_predictions(sd, tab, x, cond, time, float(w))
        if time_next < 0:
            x = x_start
            continue
        alpha, alpha_next = ac[time], ac[time_next]
        sigma = 1 * ((1 - alpha / alpha_next) * (1 - alpha_next) / (1 - alpha)).sqrt()
        c = (1 - alpha_next - sigma ** 2).sqrt()
        x = x_start * alpha_next.sqrt() + c * pred_noise + sigma * step_noise(time, x.shape)
        if x_0 is not None:
            x = _overwrite_traj(x, x_0.reshape(shape[0], -1, 3), x_0.shape[1], nf)
        if time > 0:
            xv = x.reshape(shape[0], seq_len, shape[1] // seq_len, nf)
            xv[1:, :half] = xv[:-1, half:].clone()
            x = xv.reshape(shape[0], -1, nf)
    if x_0 is not None:
        x = _overwrite_traj(x, x_0.reshape(shape[0], -1, 3), x_0.shape[1], nf)
    return x


def q_sample(tab, x_start, t: torch.Tensor, noise: torch.Tensor):
    """model/diffusion.py:625-634."""
    sh = (-1,) + (1,) * (x_start.dim() - 1)
    return tab["sqrt_alphas_cumprod"][t].reshape(sh) * x_start + \
        tab["sqrt_one_minus_alphas_cumprod"][t].reshape(sh) * noise


# ----------------------------------------------------------------------------------------------
# synthetic, name-keyed weights: identical on every box, independent of
# module construction order.
# ----------------------------------------------------------------------------------------------
def reference_param_shapes(nfeats=151, seq_len=150, latent=512, ff=1024, n_layers=8, n_head=8,
                           cond_dim=438, dn=3, use_rotary=True) -> Dict[str, tuple]:
    """Every state_dict entry of the reference DanceDecoder (attribute names model/model.py:440-540)."""
    D = latent
    s: Dict[str, tuple] = {}

    def lin(p, o, i, bias=True):
        s[p + ".weight"] = (o, i)
        if bias:
            s[p + ".bias"] = (o,)

    def ln(p):
        s[p + ".weight"] = (D,)
        s[p + ".bias"] = (D,)

    # the single RotaryEmbedding module is registered under every layer that holds it, so its
    # buffer appears once per holder in state_dict() (model/model.py:444,483,514)
    if use_rotary:
        s["rotary.freqs"] = (D // 2,)
        for i in range(2):
            s[f"cond_encoder.{i}.rotary.freqs"] = (D // 2,)
        for i in range(n_layers):
            s[f"seqTransDecoder.stack.{i}.rotary.freqs"] = (D // 2,)
    else:
        s["abs_pos_encoding.pe"] = (500, 1, D)      # PositionalEncoding's buffer (model/utils.py:12,18-25)
    lin("time_mlp.1", 4 * D, D)
    lin("to_time_cond.0", D, 4 * D)
    lin("to_time_tokens.0", 2 * D, 4 * D)
    s["null_cond_embed"] = (1, seq_len, D)
    s["null_cond_hidden"] = (1, D)
    ln("norm_cond")
    lin("input_projection", D, nfeats)
    for i in range(2):
        p = f"cond_encoder.{i}"
        s[p + ".self_attn.in_proj_weight"] = (3 * D, D)
        s[p + ".self_attn.in_proj_bias"] = (3 * D,)
        lin(p + ".self_attn.out_proj", D, D)
        lin(p + ".linear1", ff, D)
        lin(p + ".linear2", D, ff)
        ln(p + ".norm1")
        ln(p + ".norm2")
    lin("cond_projection.0", cond_dim, 2 * cond_dim)
    lin("cond_projection.2", D, cond_dim)
    ln("non_attn_cond_projection.0")
    lin("non_attn_cond_projection.1", D, D)
    lin("non_attn_cond_projection.3", D, D)
    for i in range(n_layers):
        p = f"seqTransDecoder.stack.{i}"
        for a in ("self_attn", "multihead_attn"):
            for w in ("w_qs", "w_ks", "w_vs"):
                lin(f"{p}.{a}.{w}", n_head * DK, D, bias=False)
            lin(f"{p}.{a}.fc", D, n_head * DK, bias=False)
            ln(f"{p}.{a}.layer_norm")
        lin(p + ".linear1", ff, D)
        lin(p + ".linear2", D, ff)
        for n in ("norm1", "norm2", "norm3", "norm4"):
            ln(f"{p}.{n}")
        for fl in ("film1", "film2", "film3"):
            lin(f"{p}.{fl}.block.1", 2 * D, D)
        lin(p + ".linear3", D, D)
        dims = [(D, 128), (128, 128), (128, D)]
        for j, (di, do) in enumerate(dims):  # dead w.r.t. the output, kept for checkpoint parity
            lin(f"{p}.traj_Modulation.{j}._layer", do, di)
            lin(f"{p}.traj_Modulation.{j}._hyper_bias", do, 512, bias=False)
            lin(f"{p}.traj_Modulation.{j}._hyper_gate", do, 512)
    lin("final_layer", nfeats, D)
    lin("relative_projection_layer.0", 2 * D, D * dn)
    lin("relative_projection_layer.2", 2 * D, 2 * D)
    lin("relative_projection_layer.4", D * dn, 2 * D)
    s["embeddings_table.weight"] = (10, DK * n_head)
    lin("traj_embedding.0", 64, 2)
    lin("traj_embedding.2", D, 64)
    return s


def synth_tensor(name: str, shape: tuple) -> torch.Tensor:
    """Name-seeded synthetic parameter."""
    import zlib
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    if name.endswith("rotary.freqs"):
        d = shape[0] * 2
        return 1.0 / (10000 ** (torch.arange(0, d, 2)[: d // 2].float() / d))
    if name == "abs_pos_encoding.pe":       # the buffer PositionalEncoding computes (model/utils.py:18-25), not a random tensor
        n, _, d = shape
        pe = torch.zeros(n, d)
        pos = torch.arange(0, n).unsqueeze(1)
        div = torch.exp(torch.arange(0, d, 2) * (-math.log(10000.0) / d))
        pe[:, 0::2] = torch.sin(pos * div)
        pe[:, 1::2] = torch.cos(pos * div)
        return pe.unsqueeze(1)
    is_norm = ".norm" in name or "layer_norm" in name or name.startswith("norm_cond") \
        or name.startswith("non_attn_cond_projection.0")
    if is_norm and name.endswith(".weight"):
        return 1.0 + 0.1 * (torch.rand(shape, generator=g) * 2 - 1)
    if (is_norm and name.endswith(".bias")) or name.startswith("null_cond"):
        return 0.1 * torch.randn(shape, generator=g)
    if name.endswith(".bias") or name.endswith("in_proj_bias"):
        # bias ~ U(-1/sqrt(fan_in), ..) needs fan_in of the matching weight: use shape-independent 1/sqrt(512)
        return (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(512.0)
    fan_in = shape[-1]
    return (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)


def synth_state_dict(**cfg) -> SD:
    return {k: synth_tensor(k, v) for k, v in sorted(reference_param_shapes(**cfg).items())}


def synth_cond(clip_idx: int, seq_len: int = 150, cond_dim: int = 438) -> torch.Tensor:
    g = torch.Generator().manual_seed(1000 + clip_idx)
    return torch.randn(2 * seq_len + 1, cond_dim, generator=g)


def synth_xT(clip_idx: int, L: int, nfeats: int = 151) -> torch.Tensor:
    g = torch.Generator().manual_seed(2000 + clip_idx)
    return torch.randn(L, nfeats, generator=g)


def synth_step_eps(clip_idx: int, step: int, L: int, nfeats: int = 151) -> torch.Tensor:
    g = torch.Generator().manual_seed((3000 + clip_idx) * 100003 + step)
    return torch.randn(L, nfeats, generator=g)


def synth_traj(clip_idx: int, L: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(4000 + clip_idx)
    return torch.rand(L, 3, generator=g) * 2 - 1


def synth_motion(clip_idx: int, L: int, nfeats: int = 151) -> torch.Tensor:
    """A full motion tensor in [-1, 1] (the x_0 of ddim_sample_Footwork, the `value` of the in-painting loops)."""
    g = torch.Generator().manual_seed(5000 + clip_idx)
    return torch.rand(L, nfeats, generator=g) * 2 - 1


def synth_inpaint_mask(L: int, nfeats: int = 151) -> torch.Tensor:
    """Constraint mask of the in-painting tests: trajectory channels everywhere plus the first sixth of the tokens."""
    m = torch.zeros(L, nfeats)
    m[:, 4:6] = 1.0
    m[: L // 6] = 1.0
    return m


def synth_q_eps(clip_idx: int, step: int, L: int, nfeats: int = 151) -> torch.Tensor:
    g = torch.Generator().manual_seed((6000 + clip_idx) * 100003 + step)
    return torch.randn(L, nfeats, generator=g)


def batch_step_noise(clip_ids, L: int, nfeats: int = 151) -> NoiseFn:
    def fn(i, shape):
        return torch.stack([synth_step_eps(c, i, L, nfeats) for c in clip_ids])
    return fn


# ======================================================================================================================
# Training-side rows: q_sample + p_losses terms, SMPL forward kinematics, Adan.
#
# Parity status of THIS block:
#   * q_sample, the reconstruction and velocity terms of p_losses and the Adan update are PINNED: golden vectors made
#     by tests/golden/make_golden_train.py from the real reference (model/diffusion.py:625-682, model/adan.py:33-123).
#   * the rotation conversions and SMPL forward kinematics are "parity unpinned": their arithmetic lives in
#     pytorch3d==0.7.1 (requirements.txt:70), which is neither installed here nor vendored in the reference.  The five
#     functions below restate pytorch3d's published definitions (real-first wxyz quaternions; Gram-Schmidt 6-D -> matrix
#     with ROWS b1, b2, b3; small-angle Taylor 0.5 - angle^2 / 48) and are cross-checked against
#     scipy.spatial.transform.Rotation in tests/test_train_cpu.py.  The reference call sites they serve:
#     dataset/quaternion.py:28-32 (ax_from_6v), vis.py:358-406 (SMPLSkeleton.forward), model/diffusion.py:693-733.
# ======================================================================================================================
SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]   # vis.py:48-73
SMPL_OFFSETS = [                                                                                      # vis.py:76-101
    [0.0, 0.0, 0.0], [0.05858135, -0.08228004, -0.01766408], [-0.06030973, -0.09051332, -0.01354254],
    [0.00443945, 0.12440352, -0.03838522], [0.04345142, -0.38646945, 0.008037],
    [-0.04325663, -0.38368791, -0.00484304], [0.00448844, 0.1379564, 0.02682033],
    [-0.01479032, -0.42687458, -0.037428], [0.01905555, -0.4200455, -0.03456167],
    [-0.00226458, 0.05603239, 0.00285505], [0.04105436, -0.06028581, 0.12204243],
    [-0.03483987, -0.06210566, 0.13032329], [-0.0133902, 0.21163553, -0.03346758],
    [0.07170245, 0.11399969, -0.01889817], [-0.08295366, 0.11247234, -0.02370739],
    [0.01011321, 0.08893734, 0.05040987], [0.12292141, 0.04520509, -0.019046],
    [-0.11322832, 0.04685326, -0.00847207], [0.2553319, -0.01564902, -0.02294649],
    [-0.26012748, -0.01436928, -0.03126873], [0.26570925, 0.01269811, -0.00737473],
    [-0.26910836, 0.00679372, -0.00602676], [0.08669055, -0.01063603, -0.01559429],
    [-0.0887537, -0.00865157, -0.01010708]]
FOOT_IDX = [7, 8, 10, 11]            # model/diffusion.py:720


def rotation_6d_to_matrix(d6: torch.Tensor) -> torch.Tensor:
    """pytorch3d.transforms.rotation_6d_to_matrix (Zhou et al. 2019): Gram-Schmidt, rows b1, b2, b3."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = F.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def _sqrt_positive_part(x: torch.Tensor) -> torch.Tensor:
    """pytorch3d 0.7.1: sqrt by masked assignment -- zero where x <= 0 and, unlike a torch.where over sqrt(clamp(x)),
    a clean ZERO subgradient there (where's unselected sqrt branch would contribute 0 * inf = NaN to autograd)."""
    ret = torch.zeros_like(x)
    m = x > 0
    ret[m] = torch.sqrt(x[m])
    return ret


def matrix_to_quaternion(matrix: torch.Tensor) -> torch.Tensor:
    """pytorch3d.transforms.matrix_to_quaternion (0.7.x): the candidate with the largest |component| is selected."""
    batch = matrix.shape[:-2]
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = torch.unbind(matrix.reshape(batch + (9,)), dim=-1)
    q_abs = _sqrt_positive_part(torch.stack([1.0 + m00 + m11 + m22, 1.0 + m00 - m11 - m22, 1.0 - m00 + m11 - m22,
                                             1.0 - m00 - m11 + m22], dim=-1))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, m21 - m12, m02 - m20, m10 - m01], dim=-1),
        torch.stack([m21 - m12, q_abs[..., 1] ** 2, m10 + m01, m02 + m20], dim=-1),
        torch.stack([m02 - m20, m10 + m01, q_abs[..., 2] ** 2, m12 + m21], dim=-1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, q_abs[..., 3] ** 2], dim=-1)], dim=-2)
    cand = cand / (2.0 * q_abs[..., None].clamp(min=0.1))
    idx = q_abs.argmax(dim=-1)
    return torch.gather(cand, -2, idx[..., None, None].expand(batch + (1, 4))).squeeze(-2)


def quaternion_to_axis_angle(q: torch.Tensor) -> torch.Tensor:
    norms = torch.norm(q[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    ang = 2 * half
    small = ang.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(ang), ang)
    k = torch.where(small, 0.5 - (ang * ang) / 48, torch.sin(half) / safe)
    return q[..., 1:] / k


def matrix_to_axis_angle(matrix: torch.Tensor) -> torch.Tensor:
    return quaternion_to_axis_angle(matrix_to_quaternion(matrix))


def ax_from_6v(q: torch.Tensor) -> torch.Tensor:
    """dataset/quaternion.py:28-32."""
    return matrix_to_axis_angle(rotation_6d_to_matrix(q))


def axis_angle_to_quaternion(aa: torch.Tensor) -> torch.Tensor:
    ang = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = ang * 0.5
    small = ang.abs() < 1e-6
    safe = torch.where(small, torch.ones_like(ang), ang)
    k = torch.where(small, 0.5 - (ang * ang) / 48, torch.sin(half) / safe)
    return torch.cat([torch.cos(half), aa * k], dim=-1)


def quaternion_raw_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    aw, ax, ay, az = torch.unbind(a, -1)
    bw, bx, by, bz = torch.unbind(b, -1)
    return torch.stack((aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx, aw * bz + ax * by - ay * bx + az * bw), -1)


def quaternion_multiply(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    ab = quaternion_raw_multiply(a, b)
    return torch.where(ab[..., 0:1] < 0, -ab, ab)          # standardize: non-negative real part


def quaternion_apply(q: torch.Tensor, point: torch.Tensor) -> torch.Tensor:
    p = torch.cat((point.new_zeros(point.shape[:-1] + (1,)), point), -1)
    inv = q * q.new_tensor([1, -1, -1, -1])
    return quaternion_raw_multiply(quaternion_raw_multiply(q, p), inv)[..., 1:]


def smpl_fk(rotations: torch.Tensor, root_positions: torch.Tensor) -> torch.Tensor:
    """SMPLSkeleton.forward (vis.py:358-406): rotations (N, L, 24, 3) axis-angle, root (N, L, 3) -> joints (N, L, 24, 3)."""
    q = axis_angle_to_quaternion(rotations)
    off = torch.tensor(SMPL_OFFSETS, dtype=rotations.dtype)
    has_children = [False] * 24
    for p in SMPL_PARENTS:
        if p != -1:
            has_children[p] = True
    pos, rot = [], []
    for i, p in enumerate(SMPL_PARENTS):
        if p == -1:
            pos.append(root_positions)
            rot.append(q[:, :, 0])
        else:
            pos.append(quaternion_apply(rot[p], off[i].expand(q.shape[0], q.shape[1], 3)) + pos[p])
            rot.append(quaternion_multiply(rot[p], q[:, :, i]) if has_children[i] else None)
    return torch.stack(pos, dim=3).permute(0, 1, 3, 2)


def p_losses(sd: SD, tab, x_start: torch.Tensor, cond: torch.Tensor, t: torch.Tensor, noise: torch.Tensor,
             keep_mask: torch.Tensor, loss_type: str = "l2", with_fk: bool = True, model_out: Optional[torch.Tensor] = None,
             drop=_nodrop):
    """model/diffusion.py:636-741 with the random draws injected: x_start (b, dn, S, C) dataset layout, noise in the
    PERMUTED layout (b, S, dn, C) as the reference draws it, keep_mask (b,) bool = prob_mask_like's result (eval mode:
    Dropout is the identity).  Returns (total, (recon, velocity, fk, foot)) with the reference's weights."""
    lf = (lambda a, b: (a - b) ** 2) if loss_type == "l2" else (lambda a, b: (a - b).abs())
    bs, dn, sq, c = x_start.shape
    xs = x_start.permute(0, 2, 1, 3)
    x_noisy = q_sample(tab, xs, t, noise).clone()
    x_noisy[:, :, :, [4, 5]] = xs[:, :, :, [4, 5]]
    x_noisy = x_noisy.reshape(bs, sq * dn, c)
    out = decoder_forward(sd, x_noisy, cond, t, keep_mask=keep_mask, drop=drop) if model_out is None else model_out
    w = tab["p2_loss_weight"][t]
    mo, tg = out.reshape(bs, sq, dn, c), xs.reshape(bs, sq, dn, c)
    loss = lf(mo, tg).reshape(bs, -1).mean(1) * w
    mc, mo = mo[..., :4], mo[..., 4:]
    tg = tg[..., 4:]
    v_loss = lf(mo[:, 1:] - mo[:, :-1], tg[:, 1:] - tg[:, :-1]).reshape(bs, -1).mean(1) * w
    if not with_fk:
        z = torch.zeros(())
        return 0.636 * loss.mean() + 2.964 * v_loss.mean(), (0.636 * loss.mean(), 2.964 * v_loss.mean(), z, z)
    mq = ax_from_6v(mo[..., 3:].reshape(bs, sq * dn, -1, 6))
    tq = ax_from_6v(tg[..., 3:].reshape(bs, sq * dn, -1, 6))
    mxp = smpl_fk(mq, mo[..., :3].reshape(bs, sq * dn, 3))
    txp = smpl_fk(tq, tg[..., :3].reshape(bs, sq * dn, 3))
    fk_loss = lf(mxp[:, :, 1:] - mxp[:, :, 0:1], txp[:, :, 1:] - txp[:, :, 0:1]).reshape(bs, -1).mean(1) * w
    static = mc > 0.95
    feet = mxp.reshape(bs, sq, dn, 24, 3)[:, :, :, FOOT_IDX]
    fv = torch.zeros_like(feet)
    fv[:, :-1] = feet[:, 1:] - feet[:, :-1]
    fv = torch.where(static[..., None], fv, torch.zeros_like(fv))
    foot_loss = lf(fv, torch.zeros_like(fv)).reshape(bs, -1).mean(1)
    losses = (0.636 * loss.mean(), 2.964 * v_loss.mean(), 0.646 * fk_loss.mean(), 10.942 * foot_loss.mean())
    return sum(losses), losses


def _f32(x) -> np.float32:
    return np.float32(x)


def _fma32(a, b, c):
    """round32(a * b + c) with one rounding (the product of two float32 is exact in float64; the float64 sum is then
    rounded once more, which differs from a true fused multiply-add only in ~2^-29 of the cases)."""
    return (np.asarray(a, np.float32).astype(np.float64) * np.asarray(b, np.float32).astype(np.float64)
            + np.asarray(c, np.float32).astype(np.float64)).astype(np.float32)


def adan_step(p, g, state, lr=1e-3, betas=(0.02, 0.08, 0.01), eps=1e-8, weight_decay=0.0):
    """One Adan.step for one tensor (model/adan.py:33-123) on float32 numpy arrays, with torch's rounding points
    (established against torch 2.10 CPU, tests/golden/make_golden_train.py): `t.mul_(s)` / `s * t` round once;
    `t.add_(o, alpha=s)` and `t.addcmul_(a, b, value=s)` are fused multiply-adds (`fma(o, s, t)`, `fma(s * a, b, t)`);
    `lr / t` is reciprocal-then-multiply (Tensor.__rtruediv__); `x ** 2` is x * x; sqrt, reciprocal and division are
    IEEE correctly rounded (torch's CPU sqrt goes through MKL VML for large tensors and is 1 ulp off in ~0.7 % of the
    elements: the golden comparison allows for exactly that); Python scalars are cast to float32 when they meet a
    float32 tensor.  state: dict(step, m, v, n, prev_grad) (zeros at step 0).  First step: m, v, n stay zero
    (adan.py:71), so only the weight decay acts."""
    b1, b2, b3 = betas
    m, v, n, pg = state["m"], state["v"], state["n"], state["prev_grad"]
    step = state["step"]
    if step > 0:
        m = _fma32(g, _f32(b1), m * _f32(1 - b1))
        gd = g - pg
        v = _fma32(gd, _f32(b2), v * _f32(1 - b2))
        nn = g + _f32(1 - b2) * gd
        nn = nn * nn
        n = _fma32(nn, _f32(b3), n * _f32(1 - b3))
    step += 1
    cm, cv, cn = (1 / (1 - (1 - b) ** step) for b in (b1, b2, b3))
    wss = (_f32(1.0) / (np.sqrt(n * _f32(cn)) + _f32(eps))) * _f32(lr)
    upd = m * _f32(cm) + (_f32(1 - b2) * v) * _f32(cv)
    p = _fma32(_f32(-1.0) * wss, upd, p) / _f32(1 + weight_decay * lr)
    state.update(step=step, m=m, v=v, n=n, prev_grad=g.copy())
    return p
