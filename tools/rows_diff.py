"""Diagnostic: the training engine with and without the row-block GEMM (tcdiff_gemm_rows) on the same weights, inputs and
dropout seed -- every saved activation of the forward, then every parameter gradient, compared tensor by tensor (relative L2,
largest difference, rows that differ by more than bf16 noise).  python tools/rows_diff.py [--batch 2] [--p 0.1]"""
import argparse
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tcdiff_amd import DanceDecoder  # noqa: E402
from tcdiff_amd import train_engine as TE  # noqa: E402


def cmp(name, a, b, out):
    if a is None or b is None or not torch.is_tensor(a) or a.dtype not in (torch.float32, torch.bfloat16):
        return
    a, b = a.float(), b.float()
    if a.dim() == 4:                       # head-major image [B, H, Lp, 64]
        a, b = a.permute(0, 2, 1, 3).reshape(-1, a.shape[1] * 64), b.permute(0, 2, 1, 3).reshape(-1, b.shape[1] * 64)
    a2, b2 = a.reshape(-1, a.shape[-1]), b.reshape(-1, b.shape[-1])
    d = (a2 - b2).norm(dim=1)
    ref = b2.norm(dim=1) + 1e-6 * float(b2.norm()) / max(1, b2.shape[0]) ** 0.5 + 1e-30
    bad = (d / ref > 0.1).nonzero().flatten()
    out.append((float((a - b).norm() / (b.norm() + 1e-30)), name, float((a - b).abs().max()), int(bad.numel()), bad[:6].tolist()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2)
    ap.add_argument("--p", type=float, default=0.1)
    a = ap.parse_args()
    dev, dn, S = "cuda", 3, 150
    torch.manual_seed(0)
    model = DanceDecoder(nfeats=151, seq_len=S, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                         cond_feature_dim=438, activation=F.gelu, required_dancer_num=dn, compute_dtype="bf16").to(dev)
    B = a.batch
    x = torch.randn(B, S * dn, 151, device=dev)
    cond = torch.randn(B, 2 * S, 438, device=dev)
    t = torch.randint(0, 1000, (B,), device=dev)
    keep = torch.tensor([i % 2 == 0 for i in range(B)], device=dev)
    dout = torch.randn(B, S * dn, 151, device=dev)
    res = {}
    for rows in (False, True):
        TE.TrainEngine.use_rows = rows
        TE.TrainEngine.use_graphs = 0
        eng = TE.TrainEngine(model, "bf16")
        out = eng.forward(x, cond, t, keep, (7, 9), a.p)
        sv = eng.sv
        saved = {"out": out.clone()}
        for k, v in sv.items():
            if torch.is_tensor(v):
                saved[k] = v.clone()
        for li, s in enumerate(sv["layers"]):
            for k, v in s.items():
                if torch.is_tensor(v):
                    saved[f"l{li}.{k}"] = v.clone()
        eng.backward(dout)
        grads = {n: eng.g(n).clone() for n in eng.slot}
        res[rows] = (saved, grads)
        print("engine with use_rows =", rows, "rows linears:", sum(lk.use_rows for lk in eng.lins.values()))
    for title, idx in (("saved activations (forward order within a layer is x h1 r1 Q K V O z1 x2 r2 Qc Oc z2 x3 h3 a f z3 h4 z4)", 0),
                       ("parameter gradients", 1)):
        out = []
        for k in res[True][idx]:
            cmp(k, res[True][idx][k], res[False][idx][k], out)
        print(title)
        key = (lambda r: r[1]) if idx == 0 else (lambda r: -r[0])
        for r in sorted(out, key=key)[:400 if idx == 0 else 25]:
            print(f"  {r[1]:40s} rel-L2 {r[0]:.2e}  max-abs {r[2]:.2e}  rows off by > 10 %: {r[3]} {r[4] if r[3] else ''}")


if __name__ == "__main__":
    main()
