"""MI355X-native ``Adan`` -- drop-in for the reference's ``model.adan.Adan`` (model/adan.py:11-123; constructed at
TCDiff.py:110 as ``Adan(model.parameters(), lr=learning_rate, weight_decay=weight_decay)``).

Same constructor, ``param_groups`` / ``state`` layout (``step``, ``prev_grad``, ``m``, ``v``, ``n`` per parameter, so
``optimizer_state_dict`` checkpoints interchange, TCDiff.py:271) and update rule -- including the first-step quirk: while
``step == 0`` the moments stay zero, so the first call only applies the weight decay (adan.py:71,96-107).

The whole parameter list is updated by ONE launch of ``tcdiff_adan_step`` (the reference issues ~15 elementwise kernels per
tensor, 435 tensors), with the reference's rounding points (fused multiply-adds where torch's ``add_(alpha=)`` /
``addcmul_`` fuse, IEEE sqrt / reciprocal / division).  Like every other entry of this package it has no CPU fallback:
parameters that are not contiguous fp32 CUDA tensors, or a ``restart_cond`` (never passed by the reference), raise.
"""
from __future__ import annotations

import torch
from torch.optim import Optimizer

from . import _lib as L
from . import kernels as K


def exists(val):
    return val is not None


class Adan(Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.02, 0.08, 0.01), eps=1e-8, weight_decay=0, restart_cond: callable = None):
        assert len(betas) == 3
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, restart_cond=restart_cond)
        super().__init__(params, defaults)
        self._tables = {}

    def _init_state(self, p):
        state = self.state[p]
        if len(state) == 0:
            state["step"] = 0
            state["prev_grad"] = torch.zeros_like(p.grad)
            state["m"] = torch.zeros_like(p.grad)
            state["v"] = torch.zeros_like(p.grad)
            state["n"] = torch.zeros_like(p.grad)
        return state

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if exists(closure):
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            lr, (beta1, beta2, beta3) = group["lr"], group["betas"]
            weight_decay, eps, restart_cond = group["weight_decay"], group["eps"], group["restart_cond"]
            ps = [p for p in group["params"] if exists(p.grad)]
            if not ps:
                continue
            states = [self._init_state(p) for p in ps]
            step0 = states[0]["step"]
            fused = all(
                p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                and not p.grad.is_sparse and s["step"] == step0 for p, s in zip(ps, states))
            if fused:
                key = (gi,) + tuple(t.data_ptr() for p, s in zip(ps, states)
                                    for t in (p.data, p.grad, s["m"], s["v"], s["n"], s["prev_grad"]))
                if self._tables.get(gi, (None,))[0] != key:
                    tab = K.adan_chunk_table([p.data for p in ps], [p.grad for p in ps], [s["m"] for s in states],
                                             [s["v"] for s in states], [s["n"] for s in states],
                                             [s["prev_grad"] for s in states], ps[0].device)
                    self._tables[gi] = (key, tab)
                step = step0 + 1
                cm, cv, cn = (1 / (1 - (1 - b) ** step) for b in (beta1, beta2, beta3))
                flags = int(step0 == 0) | (4 if exists(restart_cond) else 0)
                sc = L.AdanScalars(beta1, 1 - beta1, beta2, 1 - beta2, beta3, 1 - beta3, cm, cv, cn, eps, lr,
                                   1 + weight_decay * lr, flags)
                K.adan_step(self._tables[gi][1], sc)
                if exists(restart_cond):
                    # adan.py:107-114: the condition sees the state as the reference leaves it at that point -- moments updated,
                    # prev_grad and step still the previous step's -- and the tensors it selects are restarted by a second launch
                    # (m = g, v = 0, n = g^2, parameter update once more); prev_grad <- grad afterwards for everybody
                    again = [(p, s) for p, s in zip(ps, states) if restart_cond(s)]
                    if again:
                        tab = K.adan_chunk_table([p.data for p, _ in again], [p.grad for p, _ in again], [s["m"] for _, s in again],
                                                 [s["v"] for _, s in again], [s["n"] for _, s in again],
                                                 [s["prev_grad"] for _, s in again], ps[0].device)
                        sc.first = 2 | 4
                        K.adan_step(tab, sc)
                    torch._foreach_copy_([s["prev_grad"] for s in states], [p.grad for p in ps])
                torch.autograd.graph.increment_version(ps)     # written through raw pointers: keep ._version honest
                for s in states:
                    s["step"] = step
                continue
            raise L.TcdiffError(
                "Adan.step runs as one fused HIP launch over contiguous fp32 CUDA parameters with equal step counts (the "
                "configuration of TCDiff.py:110); there is no CPU / per-tensor fallback")
        return loss
