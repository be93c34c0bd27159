"""Timing of the training-side kernels that exist (scope row f1/f2: forward + loss, Adan, EMA; NO backward yet) on one
MI355X, each against the roofline that bounds it.  Prints one JSON object.

    python tools/train_bench.py [--batch 4] [--iters 20]

  adan_step / ema_update : HBM-bound streaming updates over all 435 parameter tensors of the config-2/5 network.
      Algorithmic bytes per element: Adan reads p, g, m, v, n, prev_g and writes p, m, v, n, prev_g (11 x 4 B);
      EMA reads ma, cur and writes ma (3 x 4 B).  achieved = bytes / average launch time, peak 8 TB/s.
  p_losses_forward       : q_sample + one conditional evaluation of the denoiser + 6-D->axis-angle + SMPL FK (x2) + the
      four loss terms (model/diffusion.py:636-741), forward only; priced in network FLOPs (27.9 GFLOP per clip and
      evaluation at 3 dancers x 150 frames = half of SURVEY.md's 55.81 for the two-branch step) against bf16 MFMA peak.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tcdiff_amd import Adan, DanceDecoder, GaussianDiffusion  # noqa: E402
from tcdiff_amd.diffusion import EMA  # noqa: E402


def timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def kernel_ms(fn, pattern, iters=10):
    """Average DEVICE duration of the kernels whose name contains `pattern` (torch.profiler), without host overhead."""
    from torch.profiler import ProfilerActivity, profile
    fn()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
    tot, cnt = 0.0, 0
    for ev in prof.key_averages():
        if pattern in ev.key:
            dt = getattr(ev, "self_device_time_total", None)
            if dt is None:
                dt = getattr(ev, "self_cuda_time_total", 0.0)
            tot, cnt = tot + dt, cnt + ev.count
    return tot / cnt / 1e3 if cnt else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4, help="clips per GPU (config 5: 32 over 8 GPUs)")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = "cuda"
    dn, S = 3, 150
    torch.manual_seed(0)
    model = DanceDecoder(nfeats=151, seq_len=S, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                         cond_feature_dim=438, activation=F.gelu, required_dancer_num=dn, compute_dtype="bf16")
    diff = GaussianDiffusion(model, S, 151, None, schedule="cosine", n_timestep=1000, predict_epsilon=False,
                             loss_type="l2", use_p2=False, cond_drop_prob=0.25, guidance_weight=2, seq_len=S).to(dev)
    n_el = sum(p.numel() for p in model.parameters())
    out = {"params": n_el, "n_tensors": len(list(model.parameters())), "device": torch.cuda.get_device_name(0)}

    for p in model.parameters():
        p.grad = torch.randn_like(p) * 1e-3
    opt = Adan(model.parameters(), lr=5e-5, weight_decay=0.02)
    opt.step()                                             # first step: state allocation, weight decay only
    call_ms = timed(opt.step, a.iters)
    ms = kernel_ms(opt.step, "adan_step_kernel")
    by = n_el * 11 * 4
    out["adan_step"] = dict(kernel_ms=round(ms, 4), call_ms=round(call_ms, 4), algorithmic_bytes=by,
                            achieved_gbps=round(by / ms / 1e6, 1), peak_gbps=8000,
                            hbm_frac=round(by / ms / 1e6 / 8000, 4), launches=1)

    ema = EMA(0.9999)
    upd = lambda: ema.update_model_average(diff.master_model, diff.model)      # noqa: E731
    call_ms = timed(upd, a.iters)
    ms = kernel_ms(upd, "ema_update_kernel")
    by = n_el * 3 * 4
    out["ema_update"] = dict(kernel_ms=round(ms, 4), call_ms=round(call_ms, 4), algorithmic_bytes=by,
                             achieved_gbps=round(by / ms / 1e6, 1), peak_gbps=8000,
                             hbm_frac=round(by / ms / 1e6 / 8000, 4), launches=1)

    b = a.batch
    x = torch.rand(b, dn, S, 151, device=dev) * 2 - 1
    cond = torch.randn(b, 2 * S + 1, 438, device=dev)
    t = torch.randint(0, 1000, (b,), device=dev)
    keep = torch.ones(b, dtype=torch.bool, device=dev)
    model.eval()
    with torch.no_grad():
        ms = timed(lambda: diff.p_losses(x, cond, t, keep_mask=keep), max(3, a.iters // 2))
    fl = 27.9e9 * b + 1.53e9 * b                            # one conditional evaluation + the music branch per clip
    out["p_losses_forward"] = dict(batch=b, ms=round(ms, 3), clips_per_s=round(b / ms * 1e3, 1),
                                   tflops=round(fl / ms / 1e9, 1), mfma_frac=round(fl / ms / 1e9 / 2500, 4),
                                   note="forward + loss only (eval-mode arithmetic, no autograd graph, no backward)")
    print(json.dumps(out))


if __name__ == "__main__":
    t0 = time.time()
    main()
