"""Golden vector of the REAL reference for ddim_sample with clip_denoised=False (the three DDIM samplers pass
clip_x_start=self.clip_denoised, model/diffusion.py:316,409,476): this container only; needs /root/reference.

    python tests/golden/make_golden_ddim_noclip.py

  c1_ddim_noclip.npz : config-1 shape (1 clip, 2 dancers x 60 frames, T = 100), 50 DDIM steps (eta = 1), no trajectory
     in-painting (the reference's in-painting reshape hard-codes 150 frames, :395-403), every draw injected:
     noclip   clip_denoised=False      clip   clip_denoised=True (same draws: the two differ wherever the clamp acts)
"""
import os
import sys

import numpy as np
import torch

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
    model, diff = refload.build_reference(sd, dn=dn, seq_len=S, n_timestep=T)
    cond = torch.stack([O.synth_cond(0, S)])
    xT = torch.stack([O.synth_xT(0, L)])
    eps_fn = O.batch_step_noise([0], L)
    times = [a for a, _ in O.ddim_time_pairs(T)]
    out = {}
    for name, clip in (("noclip", False), ("clip", True)):
        diff.clip_denoised = clip
        calls = {"n": 0}

        def like(tensor):
            t = times[calls["n"]]
            calls["n"] += 1
            return eps_fn(t, tensor.shape)
        with torch.no_grad(), refload.patched_randn(like_fn=like, randn_fn=lambda *a, **k: xT.clone()):
            x = diff.ddim_sample((1, L, 151), cond)
        out[name] = x.numpy()
        print(name, "max |x|", float(x.abs().max()), "draws", calls["n"])
    print("max |noclip - clip|", float(np.abs(out["noclip"] - out["clip"]).max()))
    np.savez_compressed(os.path.join(HERE, "c1_ddim_noclip.npz"), **out)


if __name__ == "__main__":
    main()
