"""MI355X-native ``DanceDecoder`` -- drop-in for the reference's ``model.model.DanceDecoder``.

Same constructor, same ``forward`` / ``guided_forward`` signatures, same ``state_dict()`` keys, shapes and
parameter order (reference model/model.py:416-624), so checkpoints, ``Adan(model.parameters())`` and the EMA
zip order (TCDiff.py:110, model/diffusion.py:67-69) work unchanged.  The module tree below only declares the
parameters (torch.nn containers); the arithmetic of ``forward`` runs in hand-written gfx950 kernels through
``tcdiff_amd.engine.DenoiserEngine``.  There is no CPU / eager fallback: calling ``forward`` without the HIP
library or off-GPU raises.

Parameters that do not influence the output in the reference (``traj_Modulation``, ``traj_embedding``,
``embeddings_table``: model/model.py:346-355,371,557,82-83) are kept for checkpoint compatibility and are not
executed.
"""
from __future__ import annotations

import math
from typing import Callable, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from . import _lib as L
from . import kernels as K
from .engine import DenoiserEngine

# Bumped by torch whenever a Parameter or a submodule is registered on ANY nn.Module of the process (global registration hooks): the
# cheap signal DanceDecoder._weights_version uses to know that its cached parameter list may be stale.
_REGISTRATIONS = [0]


def _count_registration(*_args):
    _REGISTRATIONS[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_count_registration)
torch.nn.modules.module.register_module_module_registration_hook(_count_registration)


class RotaryEmbedding(nn.Module):
    """Holds the ``freqs`` buffer (model/rotary_embedding_torch.py:75-105, freqs_for='lang', theta=1e4)."""

    def __init__(self, dim: int, theta: float = 10000.0):
        super().__init__()
        self.register_buffer("freqs", 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim)))


class _NoParam(nn.Module):
    """Placeholder occupying a Sequential slot that has no parameters in the reference."""


class DenseFiLM(nn.Module):
    def __init__(self, d: int):
        super().__init__()
        self.block = nn.Sequential(nn.Mish(), nn.Linear(d, 2 * d))


class SBI_MSA(nn.Module):
    def __init__(self, n_head: int, d_model: int, dropout: float = 0.1, dk: int = 64):
        super().__init__()
        self.n_head, self.d_k = n_head, dk
        self.w_qs = nn.Linear(d_model, n_head * dk, bias=False)
        self.w_ks = nn.Linear(d_model, n_head * dk, bias=False)
        self.w_vs = nn.Linear(d_model, n_head * dk, bias=False)
        self.fc = nn.Linear(n_head * dk, d_model, bias=False)
        self.layer_norm = nn.LayerNorm(d_model, eps=1e-6)


class ConcatSquashLinear(nn.Module):
    def __init__(self, dim_in, dim_out, dim_ctx):
        super().__init__()
        self._layer = nn.Linear(dim_in, dim_out)
        self._hyper_bias = nn.Linear(dim_ctx, dim_out, bias=False)
        self._hyper_gate = nn.Linear(dim_ctx, dim_out)


class TransformerEncoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward, dropout, rotary):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout, batch_first=True)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model, eps=1e-5)
        self.norm2 = nn.LayerNorm(d_model, eps=1e-5)
        self.rotary = rotary


class FiLMTransformerDecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, dim_feedforward, dropout, rotary, context_dim=512):
        super().__init__()
        self.self_attn = SBI_MSA(nhead, d_model, dropout=dropout)
        self.multihead_attn = SBI_MSA(nhead, d_model, dropout=dropout)
        self.linear1 = nn.Linear(d_model, dim_feedforward)
        self.linear2 = nn.Linear(dim_feedforward, d_model)
        self.norm1 = nn.LayerNorm(d_model, eps=1e-5)
        self.norm2 = nn.LayerNorm(d_model, eps=1e-5)
        self.norm3 = nn.LayerNorm(d_model, eps=1e-5)
        self.film1 = DenseFiLM(d_model)
        self.film2 = DenseFiLM(d_model)
        self.film3 = DenseFiLM(d_model)
        self.rotary = rotary
        self.linear3 = nn.Linear(d_model, d_model)
        self.norm4 = nn.LayerNorm(d_model, eps=1e-5)
        self.traj_Modulation = nn.ModuleList([
            ConcatSquashLinear(d_model, 128, context_dim),
            ConcatSquashLinear(128, 128, context_dim),
            ConcatSquashLinear(128, d_model, context_dim),
        ])


class DecoderLayerStack(nn.Module):
    def __init__(self, stack):
        super().__init__()
        self.stack = stack


class PositionalEncoding(nn.Module):
    """Holds the ``pe`` buffer of model/utils.py:11-32 ([max_len = 500, 1, d_model]: sin / cos of position x 10000^(-2i/d)); the
    engine adds its rows to the tokens (eval mode: the module's dropout is the identity)."""

    def __init__(self, d_model: int, max_len: int = 500):
        super().__init__()
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(1))


class DanceDecoder(nn.Module):
    def __init__(
        self,
        nfeats: int,
        seq_len: int = 150,
        latent_dim: int = 256,
        ff_size: int = 1024,
        num_layers: int = 4,
        num_heads: int = 4,
        dropout: float = 0.1,
        cond_feature_dim: int = 4800,
        activation: Callable[[Tensor], Tensor] = F.gelu,
        use_rotary=True,
        required_dancer_num=4,
        compute_dtype: str = "bf16",
        **kwargs,
    ) -> None:
        super().__init__()
        # use_rotary=False (model/model.py:441-448; outside the production configuration, TCDiff.py:76-87): no rotation anywhere and
        # PositionalEncoding added to the motion tokens and the music tokens instead -- inference on the op-by-op kernels
        self.use_rotary = bool(use_rotary)
        # feed-forward activation of the encoder / decoder layers (model/model.py:244,400): the production configuration passes
        # F.gelu (TCDiff.py:85) and only that runs on the fused chain kernels; relu / silu / mish run on the op-by-op kernels
        acts = {F.gelu: L.ACT_GELU, F.relu: L.ACT_RELU, F.silu: L.ACT_SILU, F.mish: L.ACT_MISH}
        if isinstance(activation, str):
            activation = {"gelu": F.gelu, "relu": F.relu, "silu": F.silu, "mish": F.mish}.get(activation, activation)
        if activation not in acts:
            raise L.TcdiffError("the MI355X path implements activation in {F.gelu, F.relu, F.silu, F.mish} (model/model.py:244,400)")
        self.act_id = acts[activation]
        self.nfeats = nfeats
        self.latent_dim = latent_dim
        self.required_dancer_num = required_dancer_num
        self.seq_len = seq_len
        self.cond_feature_dim = cond_feature_dim
        self.ff_size, self.num_layers, self.num_heads = ff_size, num_layers, num_heads
        self.compute_dtype = compute_dtype
        self.dropout_p = float(dropout)
        D = latent_dim

        self.rotary = RotaryEmbedding(dim=D) if self.use_rotary else None
        self.abs_pos_encoding = nn.Identity() if self.use_rotary else PositionalEncoding(D)
        self.time_mlp = nn.Sequential(_NoParam(), nn.Linear(D, D * 4), nn.Mish())
        self.to_time_cond = nn.Sequential(nn.Linear(D * 4, D))
        self.to_time_tokens = nn.Sequential(nn.Linear(D * 4, D * 2), _NoParam())
        self.null_cond_embed = nn.Parameter(torch.randn(1, seq_len, D))
        self.null_cond_hidden = nn.Parameter(torch.randn(1, D))
        self.norm_cond = nn.LayerNorm(D)
        self.input_projection = nn.Linear(nfeats, D)
        self.cond_encoder = nn.Sequential()
        for _ in range(2):
            self.cond_encoder.append(TransformerEncoderLayer(D, num_heads, ff_size, dropout, self.rotary))
        self.cond_projection = nn.Sequential(
            nn.Linear(cond_feature_dim * 2, cond_feature_dim), nn.ReLU(), nn.Linear(cond_feature_dim, D))
        self.non_attn_cond_projection = nn.Sequential(nn.LayerNorm(D), nn.Linear(D, D), nn.SiLU(), nn.Linear(D, D))
        stack = nn.ModuleList([FiLMTransformerDecoderLayer(D, num_heads, ff_size, dropout, self.rotary)
                               for _ in range(num_layers)])
        self.seqTransDecoder = DecoderLayerStack(stack)
        self.final_layer = nn.Linear(D, nfeats)
        self.relative_projection_layer = nn.Sequential(
            nn.Linear(D * required_dancer_num, D * 2), nn.ReLU(), nn.Linear(D * 2, D * 2), nn.ReLU(),
            nn.Linear(D * 2, D * required_dancer_num))
        self.d_k = 64
        self.embeddings_table = nn.Embedding(10, self.d_k * num_heads)
        self.traj_embedding = nn.Sequential(nn.Linear(2, 64), nn.ReLU(), nn.Linear(64, D))
        self._engine: Optional[DenoiserEngine] = None
        self._engines = {}
        self._train_engine = None
        self.train_seed = None          # (int, int): inject the dropout seed of the next train-mode forward (parity tests)

    # ------------------------------------------------------------------------------------------
    # engine plumbing
    # ------------------------------------------------------------------------------------------
    def engine_config(self) -> dict:
        return dict(latent=self.latent_dim, nfeats=self.nfeats, dn=self.required_dancer_num, seq_len=self.seq_len,
                    n_layers=self.num_layers, n_head=self.num_heads, ff=self.ff_size, cond_dim=self.cond_feature_dim,
                    act=self.act_id, abs_pos=not self.use_rotary)

    def _weights_version(self):
        """What the packed weights of an engine were made from: the identity and in-place version of every Parameter, and the device.
        Walking the module tree for the 435 Parameters costs 0.3-0.8 ms of host time -- in front of every sampler call, i.e. of a 25-ms
        one-clip job -- so the LIST is kept and refreshed only when torch reports a (sub)module or Parameter registration somewhere in the
        process (_REGISTRATIONS below: `module.weight = nn.Parameter(...)`, `module.sub = ...`, `load_state_dict` does not register)."""
        if self.__dict__.get("_plist_epoch") != _REGISTRATIONS[0]:
            self.__dict__["_plist"] = list(self.parameters())
            self.__dict__["_plist_ids"] = tuple(map(id, self._plist))
            self.__dict__["_plist_epoch"] = _REGISTRATIONS[0]
        pl = self.__dict__["_plist"]
        return tuple(p._version for p in pl) + (self.__dict__["_plist_ids"], str(pl[0].device))

    def engine(self, batch: int, slot: int = 0) -> DenoiserEngine:
        """The (lazily built) kernel engine, with weights re-packed whenever a parameter changed in place.
        `slot` selects one of several independent engines (own workspaces) used by the two-stream sampler."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise L.TcdiffError("DanceDecoder.forward runs on MI355X only: move the module to cuda "
                                "(no CPU fallback; the CPU oracle lives in oracle/ and is test-only)")
        eng = self._engines.get(slot)
        if eng is None or eng.dev != dev or eng.dt != K.dtype_id(self.compute_dtype):
            eng = DenoiserEngine(self.engine_config(), dev, self.compute_dtype)
            self._engines[slot] = eng
        ver = self._weights_version()
        if eng.weights_version != ver:
            eng.load_weights(self.state_dict(), version=ver)
        eng.plan(batch)
        if slot == 0:
            self._engine = eng
        return eng

    def set_compute_dtype(self, compute_dtype: str):
        """'bf16' (throughput), 'f32' (exact-fp32 MFMA, parity mode) or 'bf16x3' (fp32 storage, split-bf16 products: the
        reference's fp32 results to well inside 1e-3 at several times the f32 mode's speed; sampler / forward only -- a training
        step in this mode runs the f32 schedule)."""
        K.dtype_id(compute_dtype)
        self.compute_dtype = compute_dtype
        self._engine = None
        self._engines = {}
        # the training engine is rebuilt by train_engine() (it compares the arithmetic mode) and hands its gradient averager on

    def train_engine(self):
        """The (lazily built) training-step engine: operand packs, flat gradient buffer, forward / backward schedule."""
        from .train_engine import TrainEngine
        eng = self._train_engine
        dev = next(self.parameters()).device
        # an opt-in gradient averager survives EVERY rebuild (new Parameter objects, another device, another arithmetic mode):
        # dropping it silently would leave the ranks training unsynchronised replicas
        sync = eng.grad_sync if eng is not None else None
        if eng is not None and eng.param_ids != tuple(id(p) for p in self.parameters()):
            # Parameter OBJECTS were replaced (load_state_dict(assign=True), a re-wrapped module): everything the engine
            # captured or cached refers to the old ones
            eng = None
        if eng is None or eng.dev != dev or eng.mode_dt != K.dtype_id(self.compute_dtype):
            eng = TrainEngine(self, self.compute_dtype)
            if sync is not None:
                eng.grad_sync = sync
            self._train_engine = eng
        return eng

    def __deepcopy__(self, memo):
        # GaussianDiffusion deep-copies the model into master_model (model/diffusion.py:101): engines hold device
        # workspaces and a back-reference to the module -- the copy builds its own
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k in ("_engine", "_train_engine") else ({} if k == "_engines" else copy.deepcopy(v, memo))
        return new

    # ------------------------------------------------------------------------------------------
    # reference API
    # ------------------------------------------------------------------------------------------
    def guided_forward(self, x, cond_embed, times, guidance_weight):
        """unc + (cond - unc) * w (model/model.py:542-546); both branches in one stacked evaluation."""
        with torch.no_grad():
            B = x.shape[0]
            x = x.reshape(B, -1, self.nfeats).float().contiguous()
            eng = self.engine(B)
            b, Lq, dev = eng.b, eng.Lseq, x.device
            tok, hid = eng.encode_music(cond_embed.to(dev))
            times = times.to(device=dev, dtype=torch.int32).reshape(-1).contiguous()
            b["hidden_all"][:B] = eng.w["null_hidden"]
            b["hidden_all"][B:2 * B] = hid
            if bool((times == times[0]).all()):
                # one timestep for the whole batch (every sampler): cache slot 0 <- null conditioning, shared by
                # all unconditional rows; slots 1..B <- clips
                eng.build_time_tables(times[:1])
                b["tidx"].zero_()
                eng.fill_kv_slots(eng.w["null_embed"], 1, 0)
                eng.fill_kv_slots(tok, B, 1)
                n_shared = B
            else:
                # per-clip timesteps: the time-token rows differ per clip, so the null slot is replicated
                eng.build_time_tables(times)
                ar = torch.arange(B, device=dev, dtype=torch.int32)
                b["tidx"][:B] = ar
                b["tidx"][B:2 * B] = ar
                eng.fill_kv_slots(eng.w["null_embed"][None].expand(B, -1, -1).contiguous(), B, 0)
                eng.fill_kv_slots(tok, B, B)
                n_shared = 0
            eng.per_step_conditioning(2 * B)
            out = eng.network(x.reshape(B * Lq, self.nfeats), B, 2, 0, n_shared, 0)
            y = torch.empty(B, Lq, self.nfeats, device=dev, dtype=torch.float32)
            K.cfg_combine(out, out[B * Lq:], 152, float(guidance_weight), y, B * Lq, self.nfeats)
            return y

    def forward(self, x: Tensor, cond_embed: Tensor, times: Tensor, cond_drop_prob: float = 0.0, trj_dist=None, *,
                keep_mask: Optional[Tensor] = None):
        """One denoiser evaluation (model/model.py:548-624).

        With gradients enabled the call records ONE autograd node whose forward and backward are the explicit HIP
        schedule of tcdiff_amd/train_engine.py: in ``.train()`` mode with the reference's dropout (probability
        ``dropout`` of the constructor, counter-hash masks keyed by a seed drawn from torch's generator or injected through
        ``self.train_seed``), in ``.eval()`` mode with dropout off.  Under ``torch.no_grad()`` it is the inference engine.
        ``trj_dist`` is accepted for signature parity and raises, as the reference does: its gathered bias [B, H, L, L]
        (model/model.py:90-97) is added to the cross-attention scores [B, H, L, S + 2] too (model/model.py:332,394), a size
        mismatch RuntimeError for every dancer count; its callers never pass it (TCDiff.py:227-229)."""
        if trj_dist is not None:
            raise L.TcdiffError("trj_dist: the reference raises for any trj_dist (its [B, H, L, L] score bias is added to the "
                                "cross-attention scores [B, H, L, S + 2] as well, model/model.py:97,394); so does this build")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .train_engine import denoiser_train
            B = x.shape[0]
            dev = next(self.parameters()).device
            self.train_engine()          # raises off-GPU (no CPU fallback)
            p_keep = 1 - cond_drop_prob
            if keep_mask is not None:
                keep = keep_mask.to(device=dev, dtype=torch.bool).reshape(B)
            elif p_keep == 1:
                keep = torch.ones(B, dtype=torch.bool, device=dev)
            elif p_keep == 0:
                keep = torch.zeros(B, dtype=torch.bool, device=dev)
            else:
                keep = torch.zeros(B, device=dev).float().uniform_(0, 1) < p_keep       # model/utils.py:52-58
            seed = self.train_seed
            if seed is None:
                seed = tuple(int(v) for v in torch.randint(0, 2 ** 31 - 1, (2,)))
            self.train_seed = None
            return denoiser_train(self, x.reshape(B, -1, self.nfeats).to(dev), cond_embed, times, keep, seed,
                                  self.dropout_p if self.training else 0.0)
        with torch.no_grad():
            B = x.shape[0]
            x = x.reshape(B, -1, self.nfeats).float().contiguous()
            eng = self.engine(B)
            b, Lq = eng.b, eng.Lseq
            dev = x.device
            # keep mask (model/utils.py:52-58): no RNG when the probability is 0 or 1
            p_keep = 1 - cond_drop_prob
            if keep_mask is not None:           # injected draw (parity tests of the training loss)
                keep = keep_mask.to(device=dev, dtype=torch.bool).reshape(B)
            elif p_keep == 1:
                keep = torch.ones(B, dtype=torch.bool, device=dev)
            elif p_keep == 0:
                keep = torch.zeros(B, dtype=torch.bool, device=dev)
            else:
                keep = torch.zeros(B, device=dev).float().uniform_(0, 1) < p_keep
            tok, hid = eng.encode_music(cond_embed.to(dev))
            # select per clip between the music conditioning and the null embeddings (model/model.py:585-589,609-610)
            sel_tok = torch.where(keep[:, None, None], tok.view(B, eng.S, 512), eng.w["null_embed"][None])
            sel_hid = torch.where(keep[:, None], hid, eng.w["null_hidden"])
            times = times.to(device=dev, dtype=torch.int32).reshape(-1).contiguous()
            eng.build_time_tables(times)
            b["hidden_all"][:B] = sel_hid
            b["tidx"][:B] = torch.arange(B, device=dev, dtype=torch.int32)
            b["tidx"][B:] = 0
            eng.fill_kv_slots(sel_tok.contiguous(), B, 0)
            eng.per_step_conditioning(B)
            out = eng.network(x.reshape(B * Lq, self.nfeats), B, 1, 0, 0, 0)
            return out[: B * Lq, : self.nfeats].reshape(B, Lq, self.nfeats).clone()
