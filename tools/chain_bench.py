"""Chain kernel duration against the number of row blocks (one block per CU): separates a per-CU limit from contention
in the shared L2 / fabric when every block streams the same weights."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcdiff_amd import _lib as L, kernels as K
from tcdiff_amd.engine import DenoiserEngine as E

dev = "cuda"
torch.manual_seed(0)
bf = torch.bfloat16
Lq, H = 450, 8
Lp = 512
def rnd(*s, scale=1.0): return (torch.randn(*s, device=dev) * scale)
W = {n: rnd(*s, scale=s[1] ** -0.5).to(bf) for n, s in [("cfc", (512, 512)), ("ff1", (1024, 512)), ("ff2", (512, 1024)), ("l3", (512, 512)), ("qkv", (1536, 512)), ("sfc", (512, 512)), ("cq", (512, 512))]}
vec = lambda base=0.0: base + 0.1 * rnd(512)
f1, f2 = E._stages_ff1(W["ff1"]), E._stages_ff2(W["ff2"])
parts = [E._stages_n512(W["cfc"])]
parts += E._ffn_order(f1, f2)
parts.append(E._stages_n512(W["l3"]))
parts += [E._stages_n512(W["qkv"][i * 512:(i + 1) * 512]) for i in range(3)]
wsB = torch.cat(parts, 1).contiguous()
wsA = torch.cat([E._stages_n512(W["sfc"]), E._stages_n512(W["cq"])], 1).contiguous()
rope = torch.empty(Lq, 512, device=dev)
K.rope_table((1.0 / (10000 ** (torch.arange(0, 512, 2).float() / 512))).to(dev), rope, Lq)
rope = K.to_cb(rope)          # the chain kernels read the table (and keep x) column-blocked
g = [vec(1), vec(), vec(1), vec(), vec(1), vec(), vec(1), vec()]
b1, b2, b3 = 0.05 * rnd(1024), vec(), vec()

def run(nblk, mode):
    M = nblk * 64
    nseq = (M + Lq - 1) // Lq
    Oa = rnd(M, 512, scale=0.5).to(bf)
    film = 0.3 * rnd(nseq, 4096)
    x = rnd(M, 512)
    Q, Kk, V = (torch.zeros(nseq, H, Lp, 64, device=dev, dtype=bf) for _ in range(3))
    def fn():
        if mode == "B":
            K.chain(L.CHAIN_B, M, Lq, Oa, wsB, mt=4, ln_eps=1e-6, film=film, film_ld=4096, xres=x, xout=x,
                    n2_g=g[2], n2_b=g[3], rope=rope, b1=b1, film3=film[:, 2048:], n4_g=g[4], n4_b=g[5], b3=b3, nn_g=g[6],
                    nn_b=g[7], q_out=Q, k_out=Kk, v_out=V, Lp=Lp, H=H)
        else:
            K.chain(L.CHAIN_A, M, Lq, Oa, wsA, mt=4, ln_eps=1e-6, film=film, film_ld=4096, xres=x, xout=x,
                    n2_g=g[2], n2_b=g[3], rope=rope, q_out=Q, Lp=Lp, H=H)
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(20): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / 20 * 1e3

for mode in ("B", "A"):
    for nblk in (1, 2, 8, 32, 64, 128, 225, 256):
        us = run(nblk, mode)
        mb = (4.5 if mode == "B" else 1.0)
        print(f"chain {mode}: {nblk:4d} blocks  {us:8.1f} us   {mb * 1e3 / us:6.1f} GB/s per CU (weights only)")
