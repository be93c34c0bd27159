import os, sys
sys.path.insert(0, "/root/repo")
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
M = 14400
for N, Kd in ((512, 512), (1024, 512)):
    dY = torch.randn(M, N, device=dev).to(bf); X = torch.randn(M, Kd, device=dev).to(bf)
    dYt = dY.t().contiguous(); Xt = X.t().contiguous()
    out = torch.zeros(N, Kd, device=dev)
    for splits in (1, 4, 16):
        a = t(lambda: K.gemm_tn(L.DT_BF16, dY, X, N, Kd, M, N, Kd, out, Kd, splits))
        b = t(lambda: K.gemm_splitk(L.DT_BF16, dYt, Xt, N, Kd, M, M, M, out, Kd, splits))
        print(f"N={N} K={Kd} splits={splits}: TN {a:7.1f} us   NT {b:7.1f} us")
# grouped: 7 problems of a decoder layer
shapes = [(512,512)]*4 + [(1024,512),(512,1024),(1024,512),(512,512)]
probs=[]
for N,Kd in shapes:
    dY = torch.randn(M, N, device=dev).to(bf); X = torch.randn(M, Kd, device=dev).to(bf)
    probs.append((dY, X, N, Kd, M, N, Kd, torch.zeros(N,Kd,device=dev), Kd))
g = t(lambda: K.gemm_tn_grouped(L.DT_BF16, probs))
fl = sum(2.0*M*N*Kd for N,Kd in shapes)
print(f"grouped layer ({len(shapes)} problems): {g:7.1f} us  {fl/g/1e6:6.0f} TFLOP/s")
g1 = t(lambda: K.gemm_tn_grouped(L.DT_BF16, probs[:1]))
print(f"grouped single 512x512: {g1:7.1f} us")
