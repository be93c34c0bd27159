"""Golden vectors of the REAL reference's Adan with a restart condition (model/adan.py:107-114): this container only.

    python tests/golden/make_golden_adan_restart.py

  adan_restart.npz : 5 steps on three tensors (sizes with vector tails), lr 5e-5, wd 0.02,
     restart_cond = lambda state: state["step"] in (1, 3) and state["m"].numel() != 37
     (the condition sees the PREVIOUS step count -- adan.py sets state["step"] after the check -- so steps 2 and 4 restart, and
     only two of the three tensors do): parameters after every step, final m / v / n / prev_grad.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import refload  # noqa: E402


def cond(state):
    return state["step"] in (1, 3) and state["m"].numel() != 37


def main():
    sys.path.insert(0, refload.REF)
    from model.adan import Adan
    sizes = [(1000,), (37,), (4, 1025)]
    g = torch.Generator().manual_seed(78)
    params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in sizes]
    opt = Adan(params, lr=5e-5, weight_decay=0.02, restart_cond=cond)
    out = {"n_steps": np.int64(5)}
    for i, p in enumerate(params):
        out[f"p{i}_init"] = p.detach().numpy().copy()
    for step in range(5):
        for i, p in enumerate(params):
            p.grad = torch.randn(p.shape, generator=g) * (0.5 + step)
            out[f"g{i}_step{step}"] = p.grad.numpy().copy()
        opt.step()
        for i, p in enumerate(params):
            out[f"p{i}_step{step}"] = p.detach().numpy().copy()
    for i, p in enumerate(params):
        st = opt.state[p]
        for k in ("m", "v", "n", "prev_grad"):
            out[f"{k}{i}_final"] = st[k].numpy().copy()
    np.savez_compressed(os.path.join(HERE, "adan_restart.npz"), **out)
    print("written", {k: v.shape for k, v in out.items() if k.endswith("_final")})


if __name__ == "__main__":
    main()
