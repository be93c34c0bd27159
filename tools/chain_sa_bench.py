"""TC_CHAIN_FULL at the benchmark's shape (32 sequences x 450 rows) in its forms, interleaved in ONE process: plain 64-row blocks
(225), blocks cut per sequence (256), + fragment-order Q / K / V outputs, + the self-attention inside the launch; beside them the
stand-alone self-attention kernel the last form replaces.

    python tools/chain_sa_bench.py [--reps 5] [--nseq 32]
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
ap.add_argument("--nseq", type=int, default=32)
ap.add_argument("--only", default="", help="substring: time only the forms whose name contains it")
a = ap.parse_args()
dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
Lq, H, Lp, S_, nkt, nseq = 450, 8, 512, 150, 5, a.nseq
M = nseq * Lq
rnd = lambda *s, scale=1.0: torch.randn(*s, device=dev) * scale
W = {n: rnd(*s, scale=s[1] ** -0.5).to(bf) for n, s in [("cfc", (512, 512)), ("ff1", (1024, 512)), ("ff2", (512, 1024)), ("l3", (512, 512)),
                                                       ("qkv", (1536, 512)), ("sfc", (512, 512)), ("cq", (512, 512))]}
vec = lambda base=0.0: base + 0.1 * rnd(512)
f1, f2 = E._stages_ff1(W["ff1"]), E._stages_ff2(W["ff2"])
parts = [E._stages_n512(W["sfc"]), E._stages_n512(W["cq"]), E._stages_n512(W["cfc"])] + E._ffn_order(f1, f2)
parts.append(E._stages_n512(W["l3"]))
parts += [E._stages_n512(W["qkv"][i * 512:(i + 1) * 512]) for i in range(3)]
ws = torch.cat(parts, 1).contiguous()
rope = torch.empty(Lq, 512, device=dev)
K.rope_table((1.0 / (10000 ** (torch.arange(0, 512, 2).float() / 512))).to(dev), rope, Lq)
rope = K.to_cb(rope)
g = [vec(1), vec(), vec(1), vec(), vec(1), vec(), vec(1), vec()]
b1, b3 = 0.05 * rnd(1024), vec()
Oa = rnd(M, 512, scale=0.5).to(bf)
film = 0.3 * rnd(nseq, 6144)
x = K.to_cb(rnd(M, 512))
kf = rnd(nseq + 1, H, nkt * 2048, scale=0.5).to(bf)
vf = rnd(nseq + 1, H, nkt * 2048, scale=0.5).to(bf)
z = lambda *s: torch.zeros(*s, device=dev, dtype=bf)
Q, Kk, V, O = z(nseq, H, Lp, 64), z(nseq, H, Lp, 64), z(nseq, H, Lp, 64), z(M, 512)
skt, nbs = (Lq + 31) // 32, (Lq + 63) // 64
qf = rnd(nseq * nbs, 8, 4, 2, 64, 8, scale=0.3).to(bf)
skf, svf = rnd(2, nseq, H, skt * 2048, scale=0.5).to(bf), rnd(2, nseq, H, skt * 2048, scale=0.5).to(bf)
Q.copy_(rnd(nseq, H, Lp, 64, scale=0.3).to(bf)); Kk.copy_(rnd(nseq, H, Lp, 64, scale=0.5).to(bf)); V.copy_(rnd(nseq, H, Lp, 64, scale=0.5).to(bf))


def launch(form):
    kw = dict(mt=4, ln_eps=1e-6, film=film, film_ld=6144, xres=x, xout=x, n2_g=g[2], n2_b=g[3], rope=rope, b1=b1, film3=film[:, 4096:],
              n4_g=g[4], n4_b=g[5], b3=b3, nn_g=g[6], nn_b=g[7], Lp=Lp, H=H, filmb=film[:, 2048:], n3_g=g[2], n3_b=g[3], kf=kf, vf=vf,
              n_shared=nseq // 2, nkt=nkt, Lk=S_ + 2)
    if form == "attention kernel":
        K.attention(L.DT_BF16, Q, Kk, V, O, nseq, H, Lq, Lq, Lp, Lp, 512)
        return
    if form in ("plain", "seq-cut"):
        kw.update(q_out=Q, k_out=Kk, v_out=V, seq_blocks=form == "seq-cut")
    else:
        kw.update(seq_blocks=True, qf_out=qf, kf_out=skf[1], vf_out=svf[1], out_nkt=skt)
        if form == "seq-cut + frag out + self-attention":
            kw.update(sa_q=qf, sa_kf=skf[0], sa_vf=svf[0], sa_nkt=skt)
    K.chain(L.CHAIN_FULL, M, Lq, Oa, ws, **kw)


forms = ["plain", "seq-cut", "seq-cut + frag out", "seq-cut + frag out + self-attention", "attention kernel"]
if a.only:
    forms = [f for f in forms if a.only in f]
times = {f: [] for f in forms}
for rep in range(a.reps):
    for f in forms:
        for _ in range(3):
            launch(f)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(20):
            launch(f)
        e.record()
        e.synchronize()
        times[f].append(s.elapsed_time(e) / 20 * 1e3)
for f in forms:
    t = sorted(times[f])
    print(f"{nseq} x {Lq} rows, {f:40s}: median {t[len(t) // 2]:7.1f} us  min {t[0]:7.1f}  max {t[-1]:7.1f}")
