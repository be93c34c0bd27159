"""diagnostic (TC_STAMP build only): per-phase timestamps of gemm_tile / gemm_rowln blocks"""
import sys, os, math, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcdiff_amd import _lib as L, kernels as K
dev, dt, T = "cuda", L.DT_BF16, torch.bfloat16
lib = L.load()
lib.tcdiff_debug_stamp_buffer.argtypes = [ctypes.c_void_p]
st = torch.zeros(4096, 8, dtype=torch.int64, device=dev)
assert lib.tcdiff_debug_stamp_buffer(st.data_ptr()) == 0
def report(label, nb, names):
    torch.cuda.synchronize()
    s = st[:nb].cpu().double() * 10.0
    t0 = s[:, 0].min()
    n = len(names)
    d = s[:, 1:n + 1] - s[:, 0:n]
    print(label, dict(zip(names, [round(float(x)) for x in d.mean(0)])), "total(ns):", round(float(s[:, n].max() - t0)))
for (M, N, Kd) in ((128, 128, 64), (14400, 512, 512), (14400, 1536, 512)):
    A = torch.randn(M, Kd, device=dev).to(T); W = (torch.randn(N, Kd, device=dev) / math.sqrt(Kd)).to(T)
    out = torch.zeros(M, N, device=dev, dtype=T)
    for _ in range(3): K.gemm_tile(dt, A, W, M, N, Kd, out=out, ldc=N)
    report(f"tile M={M} N={N} K={Kd}", ((M + 127) // 128) * ((N + 127) // 128), ["issue", "wait0", "loop", "to_lds", "stores"])
M, Lq = 14400, 450
for Kd in (512, 1024):
    A = torch.randn(M, Kd, device=dev).to(T); W = (torch.randn(512, Kd, device=dev) / math.sqrt(Kd)).to(T)
    v = lambda: torch.randn(512, device=dev)
    bias, g1, b1, g2, b2 = v(), v(), v(), v(), v()
    film = torch.randn(32, 24576, device=dev); x = torch.randn(M, 512, device=dev); xo = torch.zeros(M, 512, device=dev)
    h = torch.zeros(M, 512, device=dev, dtype=T); r = torch.zeros(M, 512, device=dev, dtype=T)
    rope = torch.randn(Lq, 512, device=dev)
    variants = {
        "bias+store_x": dict(flags=L.ROW_BIAS | L.ROW_STORE_X, bias=bias, xout=xo),
        "bias+res+store_x": dict(flags=L.ROW_BIAS | L.ROW_RES | L.ROW_STORE_X, bias=bias, xres=x, xout=xo),
        "ln+film+res+store_x": dict(flags=L.ROW_LN_POST | L.ROW_FILM | L.ROW_STORE_X, ln_g=g1, ln_b=b1, film=film, film_ld=24576, xres=x, xout=xo),
        "full(ln+film+x+nextln+rot)": dict(flags=L.ROW_LN_POST | L.ROW_FILM | L.ROW_STORE_X | L.ROW_NEXT_LN | L.ROW_STORE_ROT, ln_g=g1, ln_b=b1,
                                           film=film, film_ld=24576, xres=x, xout=xo, nln_g=g2, nln_b=b2, rout=r, rope=rope),
        "film+nextln+h (no x)": dict(flags=L.ROW_BIAS | L.ROW_FILM | L.ROW_NEXT_LN | L.ROW_STORE_H, bias=bias, film=film, film_ld=24576, xres=x,
                                     nln_g=g2, nln_b=b2, hout=h),
    }
    for name, kw in variants.items():
        for _ in range(3): K.gemm_rowln(dt, A, W, M, Kd, Lseq=Lq, **kw)
        report(f"rowln K={Kd} {name}", (M + 63) // 64, ["loop", "phase1", "phase2"])
