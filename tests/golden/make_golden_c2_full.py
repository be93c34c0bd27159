"""Golden of the FULL benchmark-length loop from the REAL reference (build container only; needs /root/reference):
one clip at the benchmark's shape (3 dancers x 150 frames), all 1000 DDPM steps of GaussianDiffusion.p_sample_loop
(model/diffusion.py:255-286) with the per-step noise injected by the keyed recipe of oracle/tcdiff_oracle.py.

    python tests/golden/make_golden_c2_full.py         # ~1 h on 8 cores; writes tests/golden/c2_p_sample_loop_full.npz

Inputs (weights, music features, x_T, per-step eps) are regenerated on any box from their name / seed keys, so only
OUTPUTS of the reference are stored: x after selected steps and the final sample.
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
torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "6")))


def main():
    t0 = time.time()
    dn, S, T = 3, 150, int(os.environ.get("GOLDEN_T", "1000"))
    L = dn * S
    sd = O.synth_state_dict(dn=dn, seq_len=S)
    model, diff = refload.build_reference(sd, dn=dn, seq_len=S, n_timestep=T)
    cond = torch.stack([O.synth_cond(0, S)])
    xT = torch.stack([O.synth_xT(0, L)])
    eps_fn = O.batch_step_noise([0], L)
    counter = {"i": T}

    def like(tensor):
        counter["i"] -= 1
        if counter["i"] % 50 == 0:
            print("step", counter["i"], "%.0f s" % (time.time() - t0), flush=True)
        return eps_fn(counter["i"], tensor.shape)

    with torch.no_grad(), refload.patched_randn(like_fn=like):
        x, chain = diff.p_sample_loop((1, L, 151), cond, noise=xT.clone(), return_diffusion=True)
    assert counter["i"] == 0
    # chain[k] is x after k steps (chain[0] = x_T): the step with timestep index T - k was just executed
    ks = sorted({k for k in (1, T // 10, T // 2, T - T // 10, T - 10, T - 1) if 1 <= k <= T})
    keep = {f"after_step_{T - k}": chain[k].numpy() for k in ks}
    np.savez_compressed(os.path.join(OUT, "c2_p_sample_loop_full.npz" if T == 1000 else f"c2_p_sample_loop_T{T}.npz"),
                        final=x.numpy(), T=np.int64(T), **keep)
    print("wrote golden after %.0f s" % (time.time() - t0), {k: v.shape for k, v in keep.items()})


if __name__ == "__main__":
    main()
