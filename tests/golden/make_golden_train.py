"""Golden vectors for the training-side rows from the REAL reference (this container only; needs /root/reference).

    python tests/golden/make_golden_train.py

  c1_p_losses.npz : GaussianDiffusion.p_losses (model/diffusion.py:636-741) at config-1 shape (b=3, 2 dancers x 60
                    frames, T=100), eval mode, with every random draw injected: t, the q_sample noise (torch.randn_like)
                    and the keep mask (model/utils.py prob_mask_like).  Stored: x_noisy fed to the model, the model
                    output, the reconstruction and velocity terms (losses[0], losses[1]).  The FK and foot terms need
                    pytorch3d (absent): ax_from_6v / SMPLSkeleton.forward are replaced by shape-only stand-ins for the
                    run and their two loss values are NOT stored.
  adan_steps.npz  : model/adan.py Adan.step, 4 steps on three tensors (sizes with vector tails), lr 5e-5, wd 0.02
                    (TCDiff.py:110, args.py:42): parameters after every step and the final state.
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


def p_losses_golden():
    dn, S, T, b = 2, 60, 100, 3
    sd = O.synth_state_dict(dn=dn, seq_len=S)
    model, diff = refload.build_reference(sd, dn=dn, seq_len=S, n_timestep=T)
    import model.diffusion as RD
    import model.model as RM

    class FakeSmpl:                      # shape-only stand-in (pytorch3d absent); its outputs are not recorded
        def forward(self, q, x):
            return torch.zeros(q.shape[0], q.shape[1], 24, 3)
    diff.smpl = FakeSmpl()
    RD.ax_from_6v = lambda q: torch.zeros(q.shape[:-1] + (3,))
    x_start = torch.stack([O.synth_motion(c, dn * S).reshape(S, dn, 151).permute(1, 0, 2) for c in range(b)])  # (b, dn, S, C)
    cond = torch.stack([O.synth_cond(c, S) for c in range(b)])
    t = torch.tensor([73, 5, 40])
    noise = torch.stack([O.synth_xT(10 + c, dn * S).reshape(S, dn, 151) for c in range(b)])     # permuted layout (b, S, dn, C)
    keep = torch.tensor([True, False, True])
    RM.prob_mask_like = lambda shape, prob, device: keep.clone()
    captured = {}
    orig_forward = model.forward

    def spy(x, cond_embed, times, cond_drop_prob=0.0, trj_dist=None):
        captured["x_noisy"] = x.detach().clone()
        out = orig_forward(x, cond_embed, times, cond_drop_prob=cond_drop_prob, trj_dist=trj_dist)
        captured["model_out"] = out.detach().clone()
        return out
    model.forward = spy
    with torch.no_grad(), refload.patched_randn(like_fn=lambda like: noise.clone()):
        total, losses = diff.p_losses(x_start, cond, t)
    np.savez_compressed(os.path.join(HERE, "c1_p_losses.npz"), t=t.numpy(), keep=keep.numpy(),
                        x_noisy=captured["x_noisy"].numpy(), model_out=captured["model_out"].numpy(),
                        recon=np.float32(losses[0].item()), velocity=np.float32(losses[1].item()))
    print("p_losses: recon", float(losses[0]), "velocity", float(losses[1]))
    # the oracle restatement against it
    tab = O.make_tables(T)
    _, ol = O.p_losses(sd, tab, x_start, cond, t, noise, keep, with_fk=False)
    print("oracle  : recon", float(ol[0]), "velocity", float(ol[1]))


def adan_golden():
    sys.path.insert(0, refload.REF)
    from model.adan import Adan
    sizes = [(1000,), (37,), (4, 1025)]
    g = torch.Generator().manual_seed(77)
    params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in sizes]
    opt = Adan(params, lr=5e-5, weight_decay=0.02)
    out = {"n_steps": np.int64(4)}
    for i, p in enumerate(params):
        out[f"p{i}_init"] = p.detach().numpy().copy()
    for step in range(4):
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
    np.savez_compressed(os.path.join(HERE, "adan_steps.npz"), **out)
    # oracle restatement, bit for bit
    for i in range(len(sizes)):
        p = out[f"p{i}_init"].copy()
        st = dict(step=0, m=np.zeros_like(p), v=np.zeros_like(p), n=np.zeros_like(p), prev_grad=np.zeros_like(p))
        for step in range(4):
            p = O.adan_step(p, out[f"g{i}_step{step}"], st, lr=5e-5, weight_decay=0.02)
            same = np.array_equal(p, out[f"p{i}_step{step}"])
            print(f"adan tensor {i} step {step}: oracle == reference: {same}  (max diff {np.abs(p - out[f'p{i}_step{step}']).max():.2e})")


if __name__ == "__main__":
    p_losses_golden()
    adan_golden()
