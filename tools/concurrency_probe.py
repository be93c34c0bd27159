"""Do two HIP streams really run half-batch kernels side by side?  N launches of one kernel on one stream against N
launches on each of two streams (different buffers), for the step's main kernels at 7200 rows."""
import sys, os, math, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcdiff_amd import _lib as L, kernels as K
dev, dt, T = "cuda", L.DT_BF16, torch.bfloat16
M, Lq, N = 7200, 450, 200

def mk():
    v = lambda *s: torch.randn(*s, device=dev)
    d = dict(A=v(M, 512).to(T), A1=v(M, 1024).to(T), W=(v(512, 512) / 22).to(T), W1=(v(512, 1024) / 32).to(T),
             Wq=(v(1536, 512) / 22).to(T), g1=v(512), b1=v(512), g2=v(512), b2=v(512), film=v(16, 24576), x=v(M, 512),
             xo=torch.zeros(M, 512, device=dev), h=torch.zeros(M, 512, device=dev, dtype=T),
             r=torch.zeros(M, 512, device=dev, dtype=T), rope=v(Lq, 512),
             Q=torch.zeros(16, 8, 512, 64, device=dev, dtype=T), Kk=torch.zeros(16, 8, 512, 64, device=dev, dtype=T),
             V=torch.zeros(16, 8, 512, 64, device=dev, dtype=T), O=torch.zeros(M, 512, device=dev, dtype=T),
             h1=torch.zeros(M, 1024, device=dev, dtype=T), W1t=(v(1024, 512) / 22).to(T),
             Kc=torch.zeros(9, 8, 256, 64, device=dev, dtype=T), Vc=torch.zeros(9, 8, 256, 64, device=dev, dtype=T))
    return d

def kernels(d):
    return {
        "gemm_rowln K=512 (113 WGs)": lambda: K.gemm_rowln(dt, d["A"], d["W"], M, 512, Lseq=Lq,
            flags=L.ROW_LN_POST | L.ROW_FILM | L.ROW_STORE_X | L.ROW_NEXT_LN | L.ROW_STORE_ROT, ln_g=d["g1"], ln_b=d["b1"],
            film=d["film"], film_ld=24576, xres=d["x"], xout=d["xo"], nln_g=d["g2"], nln_b=d["b2"], rout=d["r"], rope=d["rope"]),
        "gemm_tile qkv (684 WGs)": lambda: K.gemm_tile(dt, d["A"], d["Wq"], M, 1536, 512, A2=d["A"], split_n=1024,
            mode=L.EPI_QKV_HEADS, out=d["Q"], out_k=d["Kk"], out_v=d["V"], scale_q=0.125, Lseq=Lq, Lp=512, H=8, n_q=512, n_k=512),
        "gemm_tile q (228 WGs)": lambda: K.gemm_tile(dt, d["A"], d["W"], M, 512, 512, mode=L.EPI_QKV_HEADS, out=d["Q"],
            scale_q=0.125, Lseq=Lq, Lp=512, H=8, n_q=512, n_k=0),
        "attention self (256 WGs)": lambda: K.attention(dt, d["Q"], d["Kk"], d["V"], d["O"], 16, 8, Lq, Lq, 512, 512, 512),
        "attention cross (256 WGs, 48 KB)": lambda: K.attention(dt, d["Q"], d["Kc"], d["Vc"], d["O"], 16, 8, Lq, 152, 512, 256, 512,
                                                                  n_shared=8),
        "gemm_tile ffn1 (456 WGs)": lambda: K.gemm_tile(dt, d["A"], d["W1t"], M, 1024, 512, act=L.ACT_GELU, out=d["h1"], ldc=1024),
    }

da, db = mk(), mk()
ka, kb = kernels(da), kernels(db)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def graph_of(f, s):
    with torch.cuda.stream(s):
        for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for _ in range(N): f()
    torch.cuda.synchronize()
    with torch.cuda.stream(s): g.replay()
    torch.cuda.synchronize()
    return g

def timed(*pairs):
    t0 = time.time()
    for g, s in pairs:
        with torch.cuda.stream(s): g.replay()
    torch.cuda.synchronize()
    return (time.time() - t0) / N * 1e6

names = list(ka)
ga = {n: graph_of(ka[n], sa) for n in names}
gb = {n: graph_of(kb[n], sb) for n in names}
single = {n: timed((ga[n], sa)) for n in names}
for n in names:
    print(f"{n:36s} alone {single[n]:6.1f} us", flush=True)
print("pairs (stream A kernel | stream B kernel): time per pair, and the sum of the two alone")
for i, na in enumerate(names):
    for nb in names[i:]:
        t = timed((ga[na], sa), (gb[nb], sb))
        print(f"  {na:34s} | {nb:34s} {t:6.1f} us  (sum {single[na] + single[nb]:6.1f}, max {max(single[na], single[nb]):6.1f})"
              f"  overlap {(single[na] + single[nb]) / t:.2f}", flush=True)
