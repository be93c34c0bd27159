import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcdiff_amd import _lib as L, kernels as K
from tools.microbench import ev  # noqa
dev, dt, T = "cuda", L.DT_BF16, torch.bfloat16
def run(M, N, Kd, mode, label):
    A = torch.randn(M, Kd, device=dev).to(T); W = (torch.randn(N, Kd, device=dev) / math.sqrt(Kd)).to(T)
    out = torch.zeros(M, N, device=dev, dtype=T if mode == L.EPI_STORE_T else torch.float32)
    us = ev(lambda: K.gemm_tile(dt, A, W, M, N, Kd, out=out, ldc=N, mode=mode))
    print(f"{label} M={M} N={N} K={Kd}: {us:7.2f} us")
counter = torch.zeros(4, dtype=torch.int32, device=dev)
print("trivial kernel (step_end): %.2f us" % ev(lambda: K.step_end(counter)))
for (M, N) in ((128, 128), (128, 512), (1024, 512), (3600, 512), (14400, 512)):
    for Kd in (64, 512):
        run(M, N, Kd, L.EPI_STORE_T, "STORE_T  ")
        run(M, N, Kd, L.EPI_STORE_F32, "STORE_F32")
for M in (64, 1024, 14400):
    for Kd in (64, 512):
        A = torch.randn(M, Kd, device=dev).to(T); W = (torch.randn(512, Kd, device=dev) / math.sqrt(Kd)).to(T)
        bias = torch.zeros(512, device=dev); xo = torch.zeros(M, 512, device=dev)
        us = ev(lambda: K.gemm_rowln(dt, A, W, M, Kd, bias=bias, xout=xo, Lseq=450, flags=L.ROW_BIAS | L.ROW_STORE_X))
        print(f"rowln M={M} K={Kd}: {us:7.2f} us")
x = torch.randn(14400, 512, device=dev); g = torch.ones(512, device=dev); b = torch.zeros(512, device=dev)
h = torch.zeros(14400, 512, device=dev, dtype=T)
print("ln_rot 14400 rows: %.2f us" % ev(lambda: K.ln_rot(dt, x, 14400, g, b, 1e-5, h=h)))
