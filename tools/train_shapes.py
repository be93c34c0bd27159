"""Per-call-site device time of one training step: every launcher of tcdiff_amd.kernels used by train_engine is wrapped with
a pair of events and keyed by (launcher, shape arguments).  python tools/train_shapes.py [--batch 32] [--top 60]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from tcdiff_amd import Adan, kernels as K
from tcdiff_amd.diffusion import GaussianDiffusion
from tcdiff_amd.model import DanceDecoder

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--top", type=int, default=70)
a = ap.parse_args()
DEV, DN, S = "cuda", 3, 150
torch.manual_seed(0)
model = DanceDecoder(nfeats=151, seq_len=S, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                     cond_feature_dim=438, activation=F.gelu, required_dancer_num=DN, compute_dtype="bf16")
diff = GaussianDiffusion(model, S, 151, None, schedule="cosine", n_timestep=1000, predict_epsilon=False, loss_type="l2",
                         use_p2=False, cond_drop_prob=0.25, guidance_weight=2, seq_len=S).to(DEV)
diff.train()
optim = Adan(model.parameters(), lr=2e-4, weight_decay=0.02)
x = torch.randn(a.batch, DN, S, 151, device=DEV)
cond = torch.randn(a.batch, 2 * S + 1, 438, device=DEV)


def step():
    tot, _ = diff(x, cond)
    optim.zero_grad()
    tot.backward()
    optim.step()
    diff.ema.update_model_average(diff.master_model, diff.model)


for _ in range(3):
    step()
torch.cuda.synchronize()

rec = []
SHAPE = {"gemm_tile": lambda a_, k: ("M%d N%d K%d" % (a_[3], a_[4], a_[5])) + (" f32out" if k.get("mode", 0) == 1 else "") +
         (" m%d" % k.get("mode", 0)),
         "gemm_tn": lambda a_, k: "N%d K%d tok%d splits%d" % (a_[3], a_[4], a_[5], a_[10]),
         "gemm_splitk": lambda a_, k: "N%d K%d tok%d splits%d" % (a_[3], a_[4], a_[5], a_[10]),
         "cast_transpose": lambda a_, k: "%s rows%d cols%d%s%s" % (str(a_[1].dtype)[6:], a_[2], a_[3],
                                                                 " +dst" if k.get("dst") is not None else "",
                                                                 " +T" if k.get("dstT") is not None else ""),
         "attention_train": lambda a_, k: "seq%d Lq%d Lk%d" % (a_[6], a_[8], a_[9]),
         "attention_bwd": lambda a_, k: "seq%d Lq%d Lk%d" % (a_[13], a_[15], a_[16]),
         "row_fwd": lambda a_, k: "M%d flags%x" % (a_[1].M, a_[1].flags),
         "row_bwd": lambda a_, k: "M%d flags%x" % (a_[1].M, a_[1].flags),
         "act_drop": lambda a_, k: "rows%d cols%d" % (a_[5], a_[6]),
         "act_drop_bwd": lambda a_, k: "rows%d cols%d" % (a_[6], a_[7])}
for name in ("gemm_tile", "gemm_splitk", "gemm_tn", "cast_transpose_multi", "cast_transpose", "attention_train", "attention_bwd", "row_fwd", "row_bwd",
             "row_param_reduce", "act_drop", "act_drop_bwd", "add_rows", "select_rows", "select_rows_bwd", "pool_bwd",
             "convert_pad", "mean_pool", "loss_terms", "loss_terms_bwd", "fk_bwd", "smpl_fk", "ax_from_6v", "adan_step",
             "ema_update", "loss_total", "q_sample_traj", "sinusoidal"):
    fn = getattr(K, name)

    def wrap(fn=fn, name=name):
        def w(*args, **kw):
            try:
                key = SHAPE[name](args, kw) if name in SHAPE else ""
            except Exception as e:      # noqa: BLE001
                key = "?" + type(e).__name__
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            r = fn(*args, **kw)
            e.record()
            rec.append((name, key, s, e))
            return r
        return w
    setattr(K, name, wrap())

N = 3
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(N):
    step()
t1.record()
torch.cuda.synchronize()
wall = t0.elapsed_time(t1) / N
agg = collections.defaultdict(lambda: [0.0, 0])
for name, key, s, e in rec:
    v = agg[(name, key)]
    v[0] += s.elapsed_time(e) / N
    v[1] += 1
tot = sum(v[0] for v in agg.values())
print(f"batch {a.batch}: {wall:.2f} ms per step with event pairs; wrapped launchers {tot:.2f} ms, {len(rec) // N} launches")
by_fn = collections.defaultdict(float)
for (name, key), v in agg.items():
    by_fn[name] += v[0]
print("  ".join(f"{n} {v:.2f}" for n, v in sorted(by_fn.items(), key=lambda kv: -kv[1])))
for (name, key), v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:a.top]:
    print(f"{v[0]:7.3f} ms  {v[1] // N:4d} x {v[0] / (v[1] / N) * 1e3:7.1f} us  {name:16s} {key}")
