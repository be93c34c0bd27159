"""diagnostic (TC_STAMP build only): per-phase timestamps of gemm_tile blocks"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcdiff_amd import _lib as L, kernels as K
dev, dt, T = "cuda", L.DT_BF16, torch.bfloat16
for (M, N, Kd) in ((128, 128, 64), (128, 128, 512), (14400, 512, 512), (14400, 1536, 512)):
    A = torch.randn(M, Kd, device=dev).to(T); W = (torch.randn(N, Kd, device=dev) / math.sqrt(Kd)).to(T)
    out = torch.zeros(M, N, device=dev, dtype=T)
    nb = ((M + 127) // 128) * ((N + 127) // 128)
    st = torch.zeros(nb, 8, dtype=torch.int64, device=dev)
    for _ in range(3):
        K.gemm_tile(dt, A, W, M, N, Kd, out=out, ldc=N, out_k=st)
    torch.cuda.synchronize()
    s = st.cpu().double() * 10.0  # ns (100 MHz)
    t0 = s[:, 0].min()
    d = (s[:, 1:6] - s[:, 0:5])
    print(f"M={M} N={N} K={Kd} blocks={nb}: mean ns per phase [issue, first-tile wait, main loop, stage->lds, stores] =",
          [round(float(x)) for x in d.mean(0)], " block start spread ns:", round(float(s[:, 0].max() - t0)),
          " last end - first start ns:", round(float(s[:, 5].max() - t0)))
