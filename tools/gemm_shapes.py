"""TFLOP/s of tcdiff_gemm_tile / tcdiff_gemm_splitk on the training step's GEMM shapes (batch 32, 3 x 150: 14 400 token rows)."""
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
M = 14400
print("forward / dgrad  C[M,N] = A[M,K] W[N,K]^T")
for N, Kd in ((512, 512), (1536, 512), (1024, 512), (512, 1024), (8192, 512), (512, 4096)):
    A = torch.randn(M, Kd, device=dev).to(bf); W = torch.randn(N, Kd, device=dev).to(bf)
    for f32 in (False, True):
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else bf)
        us = t(lambda: K.gemm_tile(L.DT_BF16, A, W, M, N, Kd, mode=L.EPI_STORE_F32 if f32 else L.EPI_STORE_T, out=out, ldc=N))
        print(f"  N={N:5d} K={Kd:5d} out={'f32' if f32 else 'bf16'}: {us:7.1f} us  {2.0*M*N*Kd/us/1e6:7.1f} TFLOP/s")
print("wgrad  dW[N,K] += dY^T[N,M] X^T[K,M]^T  (split-K, fp32 atomics)")
for N, Kd in ((512, 512), (1536, 512), (1024, 512), (512, 1024)):
    A = torch.randn(N, M, device=dev).to(bf); W = torch.randn(Kd, M, device=dev).to(bf)
    out = torch.zeros(N, Kd, device=dev)
    tiles = ((N + 127) // 128) * ((Kd + 127) // 128)
    for splits in sorted({1, max(1, 256 // tiles), max(1, 512 // tiles), max(1, 1024 // tiles)}):
        us = t(lambda: K.gemm_splitk(L.DT_BF16, A, W, N, Kd, M, M, M, out, Kd, splits))
        print(f"  N={N:5d} K={Kd:5d} splits={splits:3d} ({tiles * splits:4d} WGs): {us:7.1f} us  {2.0*M*N*Kd/us/1e6:7.1f} TFLOP/s")
print("cast_transpose [M, C] bf16 -> [C, M]")
for C in (512, 1024, 1536):
    X = torch.randn(M, C, device=dev).to(bf); Xt = torch.empty(C, M, device=dev, dtype=bf)
    us = t(lambda: K.cast_transpose(L.DT_BF16, X, M, C, C, dstT=Xt, ld_dstT=M, rows_pad=M))
    print(f"  C={C:5d}: {us:6.1f} us  {2*M*C*2/us/1e3:7.1f} GB/s")

print("library reference (torch.mm -> hipBLASLt / rocBLAS), same shapes: what a tuned vendor kernel reaches here")
for N, Kd in ((512, 512), (1536, 512), (1024, 512), (512, 1024)):
    A = torch.randn(M, Kd, device=dev).to(bf); W = torch.randn(N, Kd, device=dev).to(bf)
    Wt = W.t()
    out = torch.empty(M, N, device=dev, dtype=bf)
    us = t(lambda: torch.mm(A, Wt, out=out))
    print(f"  fwd   N={N:5d} K={Kd:5d}: {us:7.1f} us  {2.0*M*N*Kd/us/1e6:7.1f} TFLOP/s")
    dY = torch.randn(M, N, device=dev).to(bf)
    dW = torch.empty(N, Kd, device=dev, dtype=bf)
    us = t(lambda: torch.mm(dY.t(), A, out=dW))
    print(f"  wgrad N={N:5d} K={Kd:5d}: {us:7.1f} us  {2.0*M*N*Kd/us/1e6:7.1f} TFLOP/s")
