"""Golden vector of the REAL reference's inpaint_loop (model/diffusion.py:519-557) with predict_epsilon=True (the reference
constructor's default; rounds 1-4 refused the combination): this container only; needs /root/reference.

    python tests/golden/make_golden_inpaint_eps.py

  c1_inpaint_eps.npz : config-1 shape (1 clip, 2 dancers x 60 frames, T = 100), the LAST 30 DDPM steps (with an epsilon-predicting
     random-weight network the early steps diverge, as in make_golden_eps.py), every draw injected: p_sample's, then q_sample's.
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
    dn, S, T, start = 2, 60, 100, 30
    L = dn * S
    sd = O.synth_state_dict(dn=dn, seq_len=S)
    _, diff = refload.build_reference(sd, dn=dn, seq_len=S, n_timestep=T)
    diff.predict_epsilon, diff.clip_denoised = True, True
    cond = torch.stack([O.synth_cond(0, S)])
    xT = torch.stack([O.synth_xT(0, L)])
    value = torch.stack([O.synth_motion(0, L)])
    mask = torch.stack([O.synth_inpaint_mask(L)])
    eps_fn = O.batch_step_noise([0], L)
    state = {"i": start - 1, "phase": 0}

    def like(tensor):          # per step i: p_sample's randn_like first, then (i > 0) q_sample's
        i = state["i"]
        if state["phase"] == 0:
            out = eps_fn(i, tensor.shape)
            if i > 0:
                state["phase"] = 1
            else:
                state["i"] -= 1
            return out
        state["phase"] = 0
        state["i"] -= 1
        return torch.stack([O.synth_q_eps(0, i, L)])
    with torch.no_grad(), refload.patched_randn(like_fn=like):
        x = diff.inpaint_loop((1, L, 151), cond, noise=xT.clone(), constraint={"mask": mask, "value": value}, start_point=start)
    assert state["i"] == -1
    np.savez_compressed(os.path.join(HERE, "c1_inpaint_eps.npz"), final=x.numpy())
    print("inpaint_loop (predict_epsilon) done; max |x|", float(x.abs().max()))


if __name__ == "__main__":
    main()
