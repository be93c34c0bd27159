"""Device time of the train-mode attention kernels (32 sequences x 8 heads, 450 x 450 and 450 x 152) with and without
dropout -- how much of the backward is the counter-hash.  python tools/attn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcdiff_amd import _lib as L, kernels as K

dev, bf = "cuda", torch.bfloat16
def t(fn, it=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it * 1e3
B, H = 32, 8
seed = torch.tensor([1, 2], dtype=torch.int32, device=dev)
for Lq, Lk in ((450, 450), (450, 152)):
    Lpq, Lpk = K.round_up(Lq, 128), K.round_up(Lk, 128)
    Q, dO = (torch.randn(B, H, Lpq, 64, device=dev).to(bf) for _ in range(2))
    Kk, V = (torch.randn(B, H, Lpk, 64, device=dev).to(bf) for _ in range(2))
    O = torch.empty(B * Lq, 512, device=dev, dtype=bf)
    lse, delta = torch.zeros(B, H, Lpq, device=dev), torch.zeros(B, H, Lpq, device=dev)
    dQ = torch.empty(B * Lq, 512, device=dev, dtype=bf)
    dK, dV = torch.empty(B * Lk, 512, device=dev, dtype=bf), torch.empty(B * Lk, 512, device=dev, dtype=bf)
    fl = 2.0 * B * H * Lq * Lk * 64 * 2
    for p in (0.0, 0.1):
        thr, sc = K.drop_params(p)
        f = t(lambda: K.attention_train(L.DT_BF16, Q, Kk, V, O, lse, B, H, Lq, Lk, Lpq, Lpk, 512, seed, 3, thr, sc))
        b = t(lambda: K.attention_bwd(L.DT_BF16, Q, Kk, V, O, dO, lse, delta, dQ, 512, dK, dV, 512, B, H, Lq, Lk, Lpq, Lpk,
                                      512, 0.125, seed, 3, thr, sc))
        print(f"Lq {Lq} Lk {Lk} p {p}: forward {f:6.1f} us ({fl / f / 1e6:5.0f} TFLOP/s)   backward {b:6.1f} us "
              f"({3.5 * fl / b / 1e6:5.0f} TFLOP/s over 7 products)")
