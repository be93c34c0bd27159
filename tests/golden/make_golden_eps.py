"""Golden vectors of the REAL reference for the constructor options outside the production configuration
(model/diffusion.py:80-95: predict_epsilon=True is the reference's DEFAULT, clip_denoised=False skips the clamp): this
container only; needs /root/reference.

    python tests/golden/make_golden_eps.py

  c1_eps.npz : config-1 shape (1 clip, 2 dancers x 60 frames, T = 100), every draw injected (oracle.batch_step_noise):
     loop_eps_clip       p_sample_loop, last 30 steps, predict_epsilon=True,  clip_denoised=True   (:176-187,230-231)
     loop_eps_noclip     the same, last 12 steps,        predict_epsilon=True,  clip_denoised=False
     loop_x0_noclip      the same, last 30 steps,        predict_epsilon=False, clip_denoised=False
     recon / velocity    p_losses (:636-682) with predict_epsilon=True: target = the injected noise (b = 3, as c1_p_losses.npz)
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
    out = {}
    for name, eps, clip, start in (("loop_eps_clip", True, True, 30), ("loop_eps_noclip", True, False, 12),
                                   ("loop_x0_noclip", False, False, 30)):
        diff.predict_epsilon, diff.clip_denoised = eps, clip
        eps_fn = O.batch_step_noise([0], L)
        counter = {"i": start}

        def like(tensor):
            counter["i"] -= 1
            return eps_fn(counter["i"], tensor.shape)
        with torch.no_grad(), refload.patched_randn(like_fn=like):
            x = diff.p_sample_loop((1, L, 151), cond, noise=xT.clone(), start_point=start)
        assert counter["i"] == 0
        out[name] = x.numpy()
        print(name, "max |x|", float(x.abs().max()))
    # ---- p_losses with predict_epsilon=True (inputs as make_golden_train.py)
    import model.diffusion as RD
    import model.model as RM
    b = 3

    class FakeSmpl:                      # shape-only stand-in (pytorch3d absent); its outputs are not recorded
        def forward(self, q, x):
            return torch.zeros(q.shape[0], q.shape[1], 24, 3)
    diff.smpl = FakeSmpl()
    RD.ax_from_6v = lambda q: torch.zeros(q.shape[:-1] + (3,))
    diff.predict_epsilon, diff.clip_denoised = True, True
    x_start = torch.stack([O.synth_motion(c, L).reshape(S, dn, 151).permute(1, 0, 2) for c in range(b)])
    condb = torch.stack([O.synth_cond(c, S) for c in range(b)])
    t = torch.tensor([73, 5, 40])
    noise = torch.stack([O.synth_xT(10 + c, L).reshape(S, dn, 151) for c in range(b)])
    keep = torch.tensor([True, False, True])
    RM.prob_mask_like = lambda shape, prob, device: keep.clone()
    with torch.no_grad(), refload.patched_randn(like_fn=lambda like: noise.clone()):
        total, losses = diff.p_losses(x_start, condb, t)
    out["recon"], out["velocity"] = np.float32(losses[0].item()), np.float32(losses[1].item())
    print("p_losses (predict_epsilon): recon", float(losses[0]), "velocity", float(losses[1]))
    np.savez_compressed(os.path.join(HERE, "c1_eps.npz"), **out)


if __name__ == "__main__":
    main()
