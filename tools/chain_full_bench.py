"""TC_CHAIN_FULL (the fused decoder-layer launch) in both wave forms -- 8 waves x 64 columns and 4 waves x 128 columns
(tcdiff_chain_args.nw) -- on the same random data, interleaved in ONE process: launch time against the number of 64-row blocks
(1 = a lone block: no power effects; 225 = the benchmark's launch), and the outputs of the two forms compared.

    python tools/chain_full_bench.py [--reps 5] [--blocks 1,64,225,256] [--mt 4]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcdiff_amd import _lib as L, kernels as K
from tcdiff_amd.engine import DenoiserEngine as E

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--blocks", default="1,64,225,256")
ap.add_argument("--mt", type=int, default=4)
ap.add_argument("--forms", default="8,4")
a = ap.parse_args()

dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
Lq, H, Lp, S_, nkt = 450, 8, 512, 150, 5


def rnd(*s, scale=1.0):
    return torch.randn(*s, device=dev) * scale


W = {n: rnd(*s, scale=s[1] ** -0.5).to(bf) for n, s in [("cfc", (512, 512)), ("ff1", (1024, 512)), ("ff2", (512, 1024)), ("l3", (512, 512)),
                                                       ("qkv", (1536, 512)), ("sfc", (512, 512)), ("cq", (512, 512))]}
vec = lambda base=0.0: base + 0.1 * rnd(512)


def stream(nw):
    f1, f2 = E._stages_ff1(W["ff1"], nw), E._stages_ff2(W["ff2"], nw)
    parts = [E._stages_n512(W["sfc"], nw), E._stages_n512(W["cq"], nw), E._stages_n512(W["cfc"], nw)] + E._ffn_order(f1, f2)
    parts.append(E._stages_n512(W["l3"], nw))
    parts += [E._stages_n512(W["qkv"][i * 512:(i + 1) * 512], nw) for i in range(3)]
    return torch.cat(parts, 1).contiguous()


forms = [int(f) for f in a.forms.split(",")]
ws = {nw: stream(nw) for nw in forms}
rope = torch.empty(Lq, 512, device=dev)
K.rope_table((1.0 / (10000 ** (torch.arange(0, 512, 2).float() / 512))).to(dev), rope, Lq)
rope = K.to_cb(rope)
g = [vec(1), vec(), vec(1), vec(), vec(1), vec(), vec(1), vec()]
b1, b3 = 0.05 * rnd(1024), vec()
rows = 16 * a.mt

for nblk in [int(b) for b in a.blocks.split(",")]:
    M = nblk * rows
    nseq = (M + Lq - 1) // Lq
    Oa = rnd(M, 512, scale=0.5).to(bf)
    film = 0.3 * rnd(nseq, 6144)
    x0 = rnd(M, 512)
    kf = rnd(nseq + 1, H, nkt * 2048, scale=0.5).to(bf)
    vf = rnd(nseq + 1, H, nkt * 2048, scale=0.5).to(bf)
    outs = {}

    def launch(nw, x, Q, Kk, V):
        K.chain(L.CHAIN_FULL, M, Lq, Oa, ws[nw], mt=a.mt, ln_eps=1e-6, film=film, film_ld=6144, xres=x, xout=x,
                n2_g=g[2], n2_b=g[3], rope=rope, b1=b1, film3=film[:, 4096:], n4_g=g[4], n4_b=g[5], b3=b3, nn_g=g[6],
                nn_b=g[7], q_out=Q, k_out=Kk, v_out=V, Lp=Lp, H=H, filmb=film[:, 2048:],
                n3_g=g[2], n3_b=g[3], kf=kf, vf=vf, n_shared=nseq // 2, nkt=nkt, Lk=S_ + 2)

    bufs = {nw: (K.to_cb(x0), *(torch.zeros(nseq, H, Lp, 64, device=dev, dtype=bf) for _ in range(3))) for nw in forms}
    for nw in forms:                                   # one launch each from the same input: the outputs must agree
        launch(nw, *bufs[nw])
    torch.cuda.synchronize()
    if len(forms) == 2:
        d = [float((bufs[forms[0]][i].float() - bufs[forms[1]][i].float()).abs().max()) for i in range(4)]
        print(f"{nblk:4d} blocks: max |8-wave - 4-wave|  x' {d[0]:.2e}  Q {d[1]:.2e}  K {d[2]:.2e}  V {d[3]:.2e}")
    times = {nw: [] for nw in forms}
    for rep in range(a.reps):
        for nw in forms:
            for _ in range(3):
                launch(nw, *bufs[nw])
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s.record()
            for _ in range(20):
                launch(nw, *bufs[nw])
            e.record()
            e.synchronize()
            times[nw].append(s.elapsed_time(e) / 20 * 1e3)
    for nw in forms:
        t = sorted(times[nw])
        print(f"{nblk:4d} blocks of {rows} rows, {nw} waves: median {t[len(t) // 2]:7.1f} us  min {t[0]:7.1f}  max {t[-1]:7.1f}")
