"""Synthetic, name-keyed weights and inputs: identical on every box and rank, independent of
module construction order; used by bench.py, smoke() and the tests (there is no network for checkpoints).
Every tensor is drawn from a CPU generator seeded by crc32(name) / the global clip index."""
from __future__ import annotations

import math
import zlib
from typing import Dict

import torch


def synth_tensor(name: str, shape) -> torch.Tensor:
    shape = tuple(shape)
    g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
    if name.endswith("rotary.freqs"):
        d = shape[0] * 2
        return 1.0 / (10000 ** (torch.arange(0, d, 2)[: d // 2].float() / d))
    is_norm = ".norm" in name or "layer_norm" in name or name.startswith("norm_cond") \
        or name.startswith("non_attn_cond_projection.0")
    if is_norm and name.endswith(".weight"):
        return 1.0 + 0.1 * (torch.rand(shape, generator=g) * 2 - 1)
    if (is_norm and name.endswith(".bias")) or name.startswith("null_cond"):
        return 0.1 * torch.randn(shape, generator=g)
    if name.endswith(".bias") or name.endswith("in_proj_bias"):
        return (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(512.0)
    return (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(shape[-1])


def synth_state_dict_like(model: torch.nn.Module) -> Dict[str, torch.Tensor]:
    return {k: synth_tensor(k, v.shape) for k, v in sorted(model.state_dict().items())}


def synth_cond(clip_idx: int, seq_len: int = 150, cond_dim: int = 438) -> torch.Tensor:
    g = torch.Generator().manual_seed(1000 + clip_idx)
    return torch.randn(2 * seq_len + 1, cond_dim, generator=g)


def synth_xT(clip_idx: int, L: int, nfeats: int = 151) -> torch.Tensor:
    g = torch.Generator().manual_seed(2000 + clip_idx)
    return torch.randn(L, nfeats, generator=g)
