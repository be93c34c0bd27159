"""Device time of the sampler's self-attention launch (32 sequences x 8 heads, 450 x 450, bf16: the benchmark's shape) and
of the encoder-sized one.  python tools/attn_infer_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
H = 8
for B, Lq, Lk in [(b, 450, 450) for b in (int(v) for v in (sys.argv[1:] or ['32', '16']))]:
    Lpq, Lpk = K.round_up(Lq, 128) + int(os.environ.get("PADQ", "0")), K.round_up(Lk, 128) + int(os.environ.get("PADK", "0"))
    Q = torch.randn(B, H, Lpq, 64, device=dev).to(bf)
    Kk, V = (torch.randn(B, H, Lpk, 64, device=dev).to(bf) for _ in range(2))
    O = torch.empty(B * Lq, 512, device=dev, dtype=bf)
    us = t(lambda: K.attention(L.DT_BF16, Q, Kk, V, O, B, H, Lq, Lk, Lpq, Lpk, 512))
    fl = 2.0 * B * H * Lq * Lk * 64 * 2
    print(f"{os.environ.get('TCDIFF_LIB_PATH', 'default')}: B {B} ({B * H} workgroups, Lp_q {Lpq} Lp_k {Lpk}) Lq {Lq} Lk {Lk}: {us:6.1f} us ({fl / us / 1e6:5.0f} TFLOP/s)")
