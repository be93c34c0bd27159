"""Golden vectors for BASELINE config 4 (5 dancers x 300 frames: L = 1500 tokens, 302 memory rows) from the REAL
reference (build container only; needs /root/reference):

    python tests/golden/make_golden_c4.py        # writes tests/golden/c4_steps.npz (~2 min on 8 cores)

one guided evaluation at t = 500 and x after the DDPM steps 999 and 998 from x_T with injected noise, one clip.
Inputs are regenerated from their name / seed keys (oracle/tcdiff_oracle.py); only reference OUTPUTS are stored.
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
    dn, S, T = 5, 300, 1000
    L = dn * S
    sd = O.synth_state_dict(dn=dn, seq_len=S)
    model, diff = refload.build_reference(sd, dn=dn, seq_len=S, n_timestep=T)
    cond = torch.stack([O.synth_cond(0, S)])
    xT = torch.stack([O.synth_xT(0, L)])
    eps_fn = O.batch_step_noise([0], L)
    out = {}
    with torch.no_grad():
        out["guided_w2_t500"] = model.guided_forward(xT, cond, torch.full((1,), 500, dtype=torch.long), 2).numpy()
        counter = {"i": T}

        def like(tensor):
            counter["i"] -= 1
            return eps_fn(counter["i"], tensor.shape)

        x = xT.clone()
        with refload.patched_randn(like_fn=like):
            for i in (999, 998):
                x, _ = diff.p_sample(x, cond, torch.full((1,), i, dtype=torch.long))
                out[f"after_step_{i}"] = x.numpy()
    np.savez(os.path.join(OUT, "c4_steps.npz"), **out)
    print("wrote c4_steps in %.0f s" % (time.time() - t0), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
