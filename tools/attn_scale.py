import os, sys
sys.path.insert(0, os.getcwd())
import torch
from tcdiff_amd import _lib as L, kernels as K
dev, bf = "cuda", torch.bfloat16
def t(fn, it=50):
    for _ in range(5): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it * 1e3
H, Lq, Lk = 8, 450, 450
for B in (1, 2, 4, 8, 16, 24, 32):
    Q = torch.randn(B, H, 512, 64, device=dev).to(bf)
    Kk, V = (torch.randn(B, H, 512, 64, device=dev).to(bf) for _ in range(2))
    O = torch.empty(B * Lq, 512, device=dev, dtype=bf)
    us = t(lambda: K.attention(L.DT_BF16, Q, Kk, V, O, B, H, Lq, Lk, 512, 512, 512, 0, 2))
    print(f"B {B:2d}: {B * H:3d} workgroups (64 query rows per wave): {us:6.1f} us per launch")
