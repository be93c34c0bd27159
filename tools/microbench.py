"""Kernel microbenchmarks on the GPU box: K / M sweeps to separate per-k-tile cost from fixed cost."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcdiff_amd import _lib as L, kernels as K

dev = "cuda"
def ev(fn, iters=20, warm=3, reps=5):
    """device time per call: `iters` launches captured in one hipGraph (no host launch gaps), replayed `reps` times"""
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): g.replay()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / (iters * reps) * 1e3  # us

def main():
    dt = L.DT_BF16
    T = torch.bfloat16
    print("== gemm_tile STORE_T: M x N x K -> us, TF/s")
    for M in (14400, 3600):
        for N in (512, 1536):
            for Kd in (64, 128, 512, 1024, 4096):
                A = torch.randn(M, Kd, device=dev).to(T); W = (torch.randn(N, Kd, device=dev) / math.sqrt(Kd)).to(T)
                out = torch.zeros(M, N, device=dev, dtype=T)
                us = ev(lambda: K.gemm_tile(dt, A, W, M, N, Kd, out=out, ldc=N))
                print(f"tile M={M} N={N} K={Kd}: {us:8.1f} us  {2.0*M*N*Kd/us/1e6:8.1f} TF/s")
    print("== gemm_rowln BIAS|STORE_X: M x K")
    for M in (14400, 7200):
        for Kd in (64, 128, 512, 1024, 2048):
            A = torch.randn(M, Kd, device=dev).to(T); W = (torch.randn(512, Kd, device=dev) / math.sqrt(Kd)).to(T)
            bias = torch.zeros(512, device=dev); xo = torch.zeros(M, 512, device=dev)
            us = ev(lambda: K.gemm_rowln(dt, A, W, M, Kd, bias=bias, xout=xo, Lseq=450, flags=L.ROW_BIAS | L.ROW_STORE_X))
            print(f"rowln M={M} K={Kd}: {us:8.1f} us  {2.0*M*512*Kd/us/1e6:8.1f} TF/s")
    print("== attention self: nseq")
    for nseq in (32, 16, 8):
        H, Lq, Lp = 8, 450, 512
        Q = torch.randn(nseq, H, Lp, 64, device=dev).to(T) * 0.3; Kk = torch.randn(nseq, H, Lp, 64, device=dev).to(T); V = torch.randn(nseq, H, Lp, 64, device=dev).to(T)
        O = torch.zeros(nseq * Lq, 512, device=dev, dtype=T)
        us = ev(lambda: K.attention(dt, Q, Kk, V, O, nseq, H, Lq, Lq, Lp, Lp, 512))
        print(f"attn nseq={nseq}: {us:8.1f} us  {4.0*nseq*H*Lq*Lq*64/us/1e6:8.1f} TF/s")


if __name__ == "__main__":
    main()
