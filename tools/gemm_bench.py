"""Time tcdiff_gemm_tile on the step's GEMM shapes (run twice with TCDIFF_GEMM_KERNEL=1 / 2 to compare kernels)."""
import os, sys, torch
sys.path.insert(0, ".")
from tcdiff_amd import _lib as L, kernels as K

PAD = int(os.environ.get("PAD", "0"))
def run(M, N, Kd, mode, act, iters=50):
    ld = Kd + PAD
    A = torch.randn(M, ld, device="cuda").bfloat16()
    A2 = torch.randn(M, ld, device="cuda").bfloat16()
    W = (torch.randn(N, ld, device="cuda") / Kd ** 0.5).bfloat16()
    if mode == L.EPI_QKV_HEADS:
        nseq = M // 450
        Q = torch.zeros(nseq, 8, 512, 64, device="cuda", dtype=torch.bfloat16)
        Kk, Vv = torch.zeros_like(Q), torch.zeros_like(Q)
        f = lambda: K.gemm_tile(L.DT_BF16, A, W, M, N, Kd, lda=ld, ldw=ld, A2=A2, split_n=1024, mode=mode, out=Q, out_k=Kk, out_v=Vv,
                                scale_q=0.125, Lseq=450, Lp=512, H=8, n_q=512, n_k=512 if N > 512 else 0)
    else:
        out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        f = lambda: K.gemm_tile(L.DT_BF16, A, W, M, N, Kd, lda=ld, ldw=ld, act=act, mode=mode, out=out, ldc=N)
    for _ in range(5): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f()
        with torch.cuda.graph(g):
            for _ in range(iters): f()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    print(f"pad={PAD} kernel={os.environ.get('TCDIFF_GEMM_KERNEL','auto')} M={M} N={N} K={Kd} mode={mode} act={act}: {us:.1f} us  "
          f"{2.0 * M * N * Kd / us * 1e-6:.0f} TFLOP/s", flush=True)

for M in (14400, 7200):
    run(M, 1536, 512, L.EPI_QKV_HEADS, 0)
    run(M, 1024, 512, L.EPI_STORE_T, L.ACT_GELU)
    run(M, 512, 512, L.EPI_QKV_HEADS, 0)
