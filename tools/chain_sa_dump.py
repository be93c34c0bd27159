"""Diagnostic: one fused-layer launch with the self-attention inside (and one without) on fixed random data; saves the outputs so that
two library builds can be compared bit for bit:  TCDIFF_LIB_PATH=.. python tools/chain_sa_dump.py out.pt [Lq nseq mt]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcdiff_amd import _lib as L, kernels as K
from tcdiff_amd.engine import DenoiserEngine as E
out = sys.argv[1]
Lq, nseq, mt = (int(v) for v in (sys.argv[2:5] if len(sys.argv) > 4 else (450, 4, 4)))
dev, bf = "cuda", torch.bfloat16
g_ = torch.Generator(device="cpu").manual_seed(5)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g_) * scale).to(dev)
H, Lp, S_ = 8, K.round_up(Lq, 128), 60
nkt = (S_ + 2 + 31) // 32
M = nseq * Lq
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
x0 = rnd(M, 512)
kf = rnd(nseq + 1, H, nkt * 2048, scale=0.5).to(bf)
vf = rnd(nseq + 1, H, nkt * 2048, scale=0.5).to(bf)
rows = 16 * mt
skt, nbs = (Lq + 31) // 32, (Lq + rows - 1) // rows
qf = rnd(nseq * nbs, 8, 4, 2, 64, 8, scale=0.6).to(bf)
skf_in, svf_in = rnd(nseq, H, skt * 2048, scale=0.7).to(bf), rnd(nseq, H, skt * 2048, scale=0.5).to(bf)
res = {}
for form in ("xatt only", "self-attention"):
    x = K.to_cb(x0)
    qo = torch.zeros_like(qf); ko = torch.zeros(nseq, H, skt * 2048, device=dev, dtype=bf); vo = torch.zeros_like(ko)
    kw = dict(mt=mt, ln_eps=1e-6, film=film, film_ld=6144, xres=x, xout=x, n2_g=g[2], n2_b=g[3], rope=rope, b1=b1, film3=film[:, 4096:],
              n4_g=g[4], n4_b=g[5], b3=b3, nn_g=g[6], nn_b=g[7], Lp=Lp, H=H, filmb=film[:, 2048:], n3_g=g[2], n3_b=g[3], kf=kf, vf=vf,
              n_shared=nseq // 2, nkt=nkt, Lk=S_ + 2, seq_blocks=True, qf_out=qo, kf_out=ko, vf_out=vo, out_nkt=skt)
    if form == "self-attention":
        kw.update(sa_q=qf, sa_kf=skf_in, sa_vf=svf_in, sa_nkt=skt)
    K.chain(L.CHAIN_FULL, M, Lq, Oa, ws, **kw)
    torch.cuda.synchronize()
    res[form] = dict(x=x.cpu(), q=qo.cpu(), k=ko.cpu(), v=vo.cpu())
torch.save(res, out)
print("saved", out, {k: float(v["x"].float().abs().mean()) for k, v in res.items()})
