"""Timing of the TRAINING STEP (BASELINE config 5's per-GPU work: batch 32, 3 dancers x 150 frames) on one MI355X:

    total, _ = diffusion(x, cond); optim.zero_grad(); total.backward(); optim.step(); ema.update_model_average(...)

(TCDiff.py:227-245) -- forward with dropout, the four-term loss, the whole backward pass, fused Adan and EMA, all HIP
launches.  Prints one JSON object: ms per step, steps/s, clips/s, the achieved FLOP rate against the algorithmic count
(3 x the forward's 27.9 GFLOP per clip for one conditional evaluation + 1.53 GFLOP of music branch, forward = 1x, backward
= 2x) and, with --kernels, the device time per kernel family (torch.profiler).

    python tools/train_bench.py [--batch 32] [--iters 10] [--compute bf16] [--kernels]
"""
import argparse
import json
import os
import sys
from collections import defaultdict

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tcdiff_amd import Adan, DanceDecoder, GaussianDiffusion  # noqa: E402


def build(compute, dn=3, S=150, dev="cuda"):
    torch.manual_seed(0)
    model = DanceDecoder(nfeats=151, seq_len=S, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                         cond_feature_dim=438, activation=F.gelu, required_dancer_num=dn, compute_dtype=compute)
    diff = GaussianDiffusion(model, S, 151, None, schedule="cosine", n_timestep=1000, predict_epsilon=False,
                             loss_type="l2", use_p2=False, cond_drop_prob=0.25, guidance_weight=2, seq_len=S).to(dev)
    diff.train()
    return model, diff


def train_step_fn(diff, optim, x, cond):
    def step():
        total, _ = diff(x, cond)
        optim.zero_grad()
        total.backward()
        optim.step()
        diff.ema.update_model_average(diff.master_model, diff.model)
        return total
    return step


def time_steps(step, iters, warm=5):      # the engine starts replaying at the third step with the same shapes
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def kernel_table(step, iters=3):
    from torch.profiler import ProfilerActivity, profile
    step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(iters):
            step()
        torch.cuda.synchronize()
    fam = defaultdict(lambda: [0.0, 0])
    for ev in prof.key_averages():
        dt = getattr(ev, "self_device_time_total", None)
        if dt is None:
            dt = getattr(ev, "self_cuda_time_total", 0.0)
        if dt <= 0:
            continue
        name = ev.key.split("<")[0].split("(")[0].replace("void ", "").strip()
        fam[name][0] += dt / iters / 1e3
        fam[name][1] += ev.count / iters
    rows = sorted(((v[0], k, v[1]) for k, v in fam.items()), reverse=True)
    tot = sum(r[0] for r in rows)
    return {"device_ms_per_step": round(tot, 3),
            "kernels": [dict(name=k, ms=round(ms, 3), launches=round(n, 1), share=round(ms / tot, 3)) for ms, k, n in rows[:16]]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--compute", default="bf16")
    ap.add_argument("--kernels", action="store_true")
    a = ap.parse_args()
    dev, dn, S, b = "cuda", 3, 150, a.batch
    model, diff = build(a.compute)
    optim = Adan(model.parameters(), lr=5e-5, weight_decay=0.02)
    x = torch.rand(b, dn, S, 151, device=dev) * 2 - 1
    cond = torch.randn(b, 2 * S + 1, 438, device=dev)
    step = train_step_fn(diff, optim, x, cond)
    ms = time_steps(step, a.iters)
    fl = 3 * (27.9e9 + 1.53e9) * b
    out = dict(config=f"train step, batch {b}, 3 dancers x 150 frames, {a.compute}, Adan + EMA, dropout 0.1", ms_per_step=round(ms, 3),
               steps_per_s=round(1e3 / ms, 2), clips_per_s=round(b * 1e3 / ms, 1), algorithmic_tflop_per_step=round(fl / 1e12, 3),
               tflops=round(fl / ms / 1e9, 1), mfma_frac=round(fl / ms / 1e9 / 2500, 4), device=torch.cuda.get_device_name(0),
               peak_mem_gb=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2))
    if a.kernels:
        out.update(kernel_table(step))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
