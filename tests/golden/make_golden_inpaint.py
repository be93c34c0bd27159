"""Golden vectors of the in-painting samplers from the REAL reference (build container only; needs /root/reference):
ddim_sample_Footwork (model/diffusion.py:289-383) at the config-2 shape, inpaint_loop (:519-557) and
long_inpaint_loop (:560-608) at the config-1 shape.  Inputs come from the seed-keyed recipes in oracle/tcdiff_oracle.py, so only the reference's OUTPUTS are stored.

    python tests/golden/make_golden_inpaint.py
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import refload  # noqa: E402
from oracle import tcdiff_oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(8)


def main():
    t0 = time.time()
    with torch.no_grad():
        # ---- ddim_sample_Footwork: 3 dancers x 150 frames, T=1000, B=1 ------------------------------------------
        dn, S, T = 3, 150, 1000
        L = dn * S
        sd = O.synth_state_dict(dn=dn, seq_len=S)
        _, diff = refload.build_reference(sd, dn=dn, seq_len=S, n_timestep=T)
        cond = torch.stack([O.synth_cond(0, S)])
        xT = torch.stack([O.synth_xT(0, L)])
        x0 = torch.stack([O.synth_motion(0, L)])
        eps_fn = O.batch_step_noise([0], L)
        times = [a for a, _ in O.ddim_time_pairs(T)]
        calls = {"n": 0}

        def like(tensor):
            t = times[calls["n"]]
            calls["n"] += 1
            return eps_fn(t, tensor.shape)

        with refload.patched_randn(like_fn=like, randn_fn=lambda *a, **k: xT.clone()):
            xf = diff.ddim_sample_Footwork((1, L, 151), cond, x_0=x0.clone())
        assert calls["n"] == 49
        np.savez(os.path.join(OUT, "c2_footwork.npz"), final=xf.numpy())
        print("footwork done", time.time() - t0, float(xf.abs().max()))

        # ---- inpaint_loop: 2 dancers x 60 frames, T=100, B=1 ------------------------------------------------------
        dn, S, T = 2, 60, 100
        L = dn * S
        sd = O.synth_state_dict(dn=dn, seq_len=S)
        _, diff = refload.build_reference(sd, dn=dn, seq_len=S, n_timestep=T)
        cond = torch.stack([O.synth_cond(0, S)])
        xT = torch.stack([O.synth_xT(0, L)])
        value = torch.stack([O.synth_motion(0, L)])
        mask = torch.stack([O.synth_inpaint_mask(L)])
        eps_fn = O.batch_step_noise([0], L)
        state = {"i": T - 1, "phase": 0}

        def like2(tensor):
            # per step i: p_sample's randn_like first, then (i > 0) q_sample's
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

        with refload.patched_randn(like_fn=like2):
            xi = diff.inpaint_loop((1, L, 151), cond, noise=xT.clone(), constraint={"mask": mask, "value": value})
        assert state["i"] == -1
        np.savez(os.path.join(OUT, "c1_inpaint.npz"), final=xi.numpy())
        print("inpaint done", time.time() - t0)

        # ---- long_inpaint_loop: same model, B=2 half-overlapping windows (model/diffusion.py:560-608) ---------------
        cond2 = torch.stack([O.synth_cond(c, S) for c in (0, 1)])
        xT2 = torch.stack([O.synth_xT(c, L) for c in (0, 1)])
        eps2 = O.batch_step_noise([0, 1], L)
        counter = {"i": T}

        def like3(tensor):
            counter["i"] -= 1
            return eps2(counter["i"], tensor.shape)

        with refload.patched_randn(like_fn=like3):
            xl = diff.long_inpaint_loop((2, L, 151), cond2, noise=xT2.clone())
        assert counter["i"] == 0
        np.savez(os.path.join(OUT, "c1_long_inpaint.npz"), final=xl.numpy())
        print("long inpaint done", time.time() - t0)


if __name__ == "__main__":
    main()
