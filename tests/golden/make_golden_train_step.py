"""Golden vectors of the TRAINING STEP from the REAL reference (this container only; needs /root/reference).

    python tests/golden/make_golden_train_step.py

  c1_train_step.npz : two consecutive steps of TCDiff.train_loop's body (TCDiff.py:227-234) on the real
        model/model.py + model/diffusion.py + model/adan.py at config-1 shape (b = 3, 2 dancers x 60 frames, T = 100,
        Adan lr 5e-5, weight decay 0.02): total_loss = diffusion.p_losses(x, cond, t); zero_grad; backward; optim.step.
        Step 0 runs in eval mode (Dropout = identity), step 1 in train mode with every dropout mask injected.
        Every random draw is injected: t, the q_sample noise, the keep mask, and the dropout masks -- the latter are the
        product's counter hash (oracle.dropout_keep == tcdiff_amd/csrc/train_common.h), fed to the reference by replacing
        nn.Dropout.forward and nn.MultiheadAttention's scaled_dot_product_attention call with mask-taking equivalents.
        Stored per step: the four loss terms, total, and for NAMED parameters (below) the gradient (small tensors in full,
        matrices as a strided sample + their L2 norm); after step 1 the same sample of the parameters themselves.
  pytorch3d is absent, so ax_from_6v / SMPLSkeleton.forward inside p_losses are the oracle's restatements (torch, hence
  differentiable): the FK and foot terms and their gradient contributions are "parity unpinned"; every other number is
  the reference's own arithmetic and torch autograd through it.
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

DN, S, T, B = 2, 60, 100, 3
LR, WD = 5e-5, 0.02
P_DROP = 0.1
STEPS = [dict(train=False, t=[73, 5, 40], keep=[True, False, True], seed=(0, 0), noise0=10),
         dict(train=True, t=[12, 88, 51], keep=[False, True, True], seed=(1234, 5678), noise0=20)]

ST = "seqTransDecoder.stack."
NAMED = [
    "input_projection.weight", "input_projection.bias", "relative_projection_layer.0.weight",
    "relative_projection_layer.4.bias", "time_mlp.1.weight", "to_time_cond.0.bias", "to_time_tokens.0.weight",
    "null_cond_embed", "null_cond_hidden", "norm_cond.weight", "cond_projection.0.weight", "cond_projection.2.bias",
    "cond_encoder.0.self_attn.in_proj_weight", "cond_encoder.0.self_attn.in_proj_bias",
    "cond_encoder.1.self_attn.out_proj.weight", "cond_encoder.1.linear1.weight", "cond_encoder.0.norm2.weight",
    "non_attn_cond_projection.0.bias", "non_attn_cond_projection.3.weight",
    ST + "0.self_attn.w_qs.weight", ST + "0.self_attn.w_vs.weight", ST + "0.self_attn.fc.weight",
    ST + "0.self_attn.layer_norm.weight", ST + "0.multihead_attn.w_ks.weight", ST + "3.multihead_attn.w_vs.weight",
    ST + "3.multihead_attn.w_qs.weight", ST + "3.norm1.weight", ST + "3.norm2.bias", ST + "3.linear1.weight",
    ST + "3.linear2.bias", ST + "3.film2.block.1.weight", ST + "7.film3.block.1.bias", ST + "7.linear3.weight",
    ST + "7.norm4.weight", ST + "7.multihead_attn.layer_norm.bias", "final_layer.weight", "final_layer.bias",
]


def sample(t: torch.Tensor) -> np.ndarray:
    f = t.detach().reshape(-1)
    stride = max(1, f.numel() // 4096)
    return f[::stride].numpy().copy()


def step_inputs(k):
    sp = STEPS[k]
    x_start = torch.stack([O.synth_motion(100 * k + c, DN * S).reshape(S, DN, 151).permute(1, 0, 2) for c in range(B)])
    cond = torch.stack([O.synth_cond(100 * k + c, S) for c in range(B)])
    noise = torch.stack([O.synth_xT(sp["noise0"] + c, DN * S).reshape(S, DN, 151) for c in range(B)])
    return x_start, cond, noise, torch.tensor(sp["t"]), torch.tensor(sp["keep"])


class patched_dropout:
    """nn.Dropout.forward and F.scaled_dot_product_attention (as called by nn.MultiheadAttention) take their masks from
    `plan`; sites as in oracle.DropPlan."""

    def __init__(self, model, plan):
        self.model, self.plan = model, plan

    def __enter__(self):
        import torch.nn as nn
        import torch.nn.functional as F
        m, plan = self.model, self.plan
        for i, lyr in enumerate(m.cond_encoder):
            lyr.dropout1._sites, lyr.dropout._sites, lyr.dropout2._sites = [4 * i + 1], [4 * i + 2], [4 * i + 3]
        for l, lyr in enumerate(m.seqTransDecoder.stack):
            sd = 16 + 8 * l
            lyr.self_attn.dropout._sites = [sd + 0, sd + 1]          # softmax weights, then the fc output (model.py:98,103)
            lyr.dropout1._sites = [sd + 2]
            lyr.multihead_attn.dropout._sites = [sd + 3, sd + 4]
            lyr.dropout2._sites = [sd + 5]
            lyr.dropout._sites = [sd + 6]
            lyr.dropout3._sites = [sd + 7]
        for mod in m.modules():
            if isinstance(mod, nn.Dropout):
                mod._k = 0
        self._orig_fwd = nn.Dropout.forward
        orig = self._orig_fwd

        def fwd(mod, x):
            if not mod.training or not hasattr(mod, "_sites"):
                return orig(mod, x)
            site = mod._sites[mod._k % len(mod._sites)]
            mod._k += 1
            return plan(x, site)
        nn.Dropout.forward = fwd
        self._orig_sdpa = F.scaled_dot_product_attention
        counter = {"k": 0}

        def sdpa(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, **kw):
            assert attn_mask is None and not is_causal
            att = torch.softmax(q @ k.transpose(-2, -1) / (q.shape[-1] ** 0.5), dim=-1)
            if dropout_p > 0.0:
                att = plan(att, 4 * (counter["k"] % 2) + 0)         # encoder layers 0, 1 in call order
                counter["k"] += 1
            return att @ v
        F.scaled_dot_product_attention = sdpa
        return self

    def __exit__(self, *exc):
        import torch.nn as nn
        import torch.nn.functional as F
        nn.Dropout.forward = self._orig_fwd
        F.scaled_dot_product_attention = self._orig_sdpa


def main():
    sd0 = O.synth_state_dict(dn=DN, seq_len=S)
    model, diff = refload.build_reference(sd0, dn=DN, seq_len=S, n_timestep=T)
    import model.diffusion as RD
    import model.model as RM
    sys.path.insert(0, refload.REF)
    from model.adan import Adan

    class OracleSmpl:                        # pytorch3d absent: the oracle's restatement (differentiable torch)
        def forward(self, q, x):
            return O.smpl_fk(q, x)
    diff.smpl = OracleSmpl()
    RD.ax_from_6v = O.ax_from_6v
    optim = Adan(model.parameters(), lr=LR, weight_decay=WD)
    out = dict(lr=np.float64(LR), wd=np.float64(WD), p_drop=np.float64(P_DROP), names=np.array(NAMED))
    tab = O.make_tables(T)
    for k, sp in enumerate(STEPS):
        x_start, cond, noise, t, keep = step_inputs(k)
        RM.prob_mask_like = lambda shape, prob, device: keep.clone()
        model.train(sp["train"])
        diff.train(sp["train"])
        plan = O.DropPlan(sp["seed"], P_DROP if sp["train"] else 0.0)
        # ---- the oracle restatement on the SAME weights / draws: pins its train mode and its autograd -----------------
        sd_now = {n: p.detach().clone().requires_grad_(True) for n, p in model.state_dict().items() if p.is_floating_point()}
        o_total, o_losses = O.p_losses(sd_now, tab, x_start, cond, t, noise, keep, drop=plan)
        o_total.backward()
        # ---- the real reference ------------------------------------------------------------------------------------------
        with patched_dropout(model, plan), refload.patched_randn(like_fn=lambda like: noise.clone()):
            total, losses = diff.p_losses(x_start, cond, t)
        optim.zero_grad()
        total.backward()
        named = dict(model.named_parameters())
        out[f"s{k}_t"], out[f"s{k}_keep"], out[f"s{k}_seed"] = t.numpy(), keep.numpy(), np.array(sp["seed"], dtype=np.int64)
        out[f"s{k}_train"] = np.bool_(sp["train"])
        out[f"s{k}_losses"] = np.array([float(v) for v in losses], dtype=np.float32)
        out[f"s{k}_total"] = np.float32(total.item())
        worst = 0.0
        for n in NAMED:
            g = named[n].grad
            out[f"s{k}_g:{n}"] = sample(g)
            out[f"s{k}_gn:{n}"] = np.float32(g.norm().item())
            og = sd_now[n].grad
            assert bool(torch.isfinite(g).all()) and bool(torch.isfinite(og).all()), n
            rel = float((og - g).norm() / (g.norm() + 1e-30))
            worst = max(worst, rel)
        dead = [n for n, p in named.items() if p.grad is None]
        out[f"s{k}_n_dead"] = np.int64(len(dead))
        print(f"step {k} ({'train' if sp['train'] else 'eval'}): total {total.item():.6f} (oracle {o_total.item():.6f}), losses "
              f"{[round(float(v), 6) for v in losses]}; worst oracle-vs-reference gradient rel-L2 over {len(NAMED)} named "
              f"parameters {worst:.2e}; {len(dead)} parameters without gradient")
        optim.step()
        for n in NAMED:
            out[f"s{k}_p:{n}"] = sample(named[n])
    np.savez_compressed(os.path.join(HERE, "c1_train_step.npz"), **out)
    print("wrote c1_train_step.npz", os.path.getsize(os.path.join(HERE, "c1_train_step.npz")) // 1024, "KB")


if __name__ == "__main__":
    main()
