"""Golden vectors of the REAL reference with a feed-forward activation other than F.gelu (model/model.py:244,400: the encoder and
decoder layers take the constructor's `activation`; TCDiff.py:85 passes F.gelu): this container only; needs /root/reference.

    python tests/golden/make_golden_activation.py

  c1_activation.npz : config-1 shape (1 clip, 2 dancers x 60 frames, T = 100); for relu / silu / mish: the guided evaluation at
     t = 50 (w = 2) and the conditional forward at t = 3.
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import refload  # noqa: E402
from oracle import tcdiff_oracle as O  # noqa: E402

torch.set_num_threads(8)


def main():
    dn, S, T = 2, 60, 100
    L = dn * S
    sd = O.synth_state_dict(dn=dn, seq_len=S)
    cond = torch.stack([O.synth_cond(0, S)])
    xT = torch.stack([O.synth_xT(0, L)])
    out = {}
    for name, fn in (("relu", F.relu), ("silu", F.silu), ("mish", F.mish)):
        model, _ = refload.build_reference(sd, dn=dn, seq_len=S, n_timestep=T, activation=fn)
        with torch.no_grad():
            out[f"{name}_guided_w2_t50"] = model.guided_forward(xT, cond, torch.tensor([50]), 2).numpy()
            out[f"{name}_fwd_cond_t3"] = model(xT, cond, torch.tensor([3]), cond_drop_prob=0.0).numpy()
        print(name, float(np.abs(out[f"{name}_guided_w2_t50"]).max()))
    np.savez_compressed(os.path.join(HERE, "c1_activation.npz"), **out)


if __name__ == "__main__":
    main()
