"""Diagnostic: tcdiff_gemm_rows against tcdiff_gemm_tile on the decoder layer's training-step products (bf16), at the token-row
counts of batch 32 and batch 4, plus the stream packer.  python tools/gemm_rows_bench.py [M ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tcdiff_amd import _lib as L  # noqa: E402
from tcdiff_amd import kernels as K  # noqa: E402

DEV, BF, DT = "cuda", torch.bfloat16, L.DT_BF16


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    Ms = [int(x) for x in sys.argv[1:]] or [28800, 14400, 3600]
    for M in Ms:
        print(f"M = {M} token rows")
        for name, N, Kd, mode in (("sfc / cfc / l3 forward (f32 out)", 512, 512, "f32"), ("cq forward / dgrad (bf16 out)", 512, 512, "t"),
                                  ("qkv forward (heads)", 1536, 512, "heads"), ("ff1 forward (GELU + dropout)", 1024, 512, "act"),
                                  ("ff2 forward (f32 out)", 512, 1024, "f32"), ("ff2 dgrad (GELU' + dropout)", 1024, 512, "actb"),
                                  ("ff1 / qk dgrad (bf16 out)", 512, 1024, "t")):
            A = torch.randn(M, Kd, device=DEV).to(BF)
            A2 = torch.randn(M, Kd, device=DEV).to(BF)
            W = torch.randn(N, Kd, device=DEV) * 0.05
            Wb = W.to(BF)
            ws = K.row_streams(W)
            bias = torch.randn(N, device=DEV)
            seed = torch.tensor([1, 2], dtype=torch.int32, device=DEV)
            thr, sc = K.drop_params(0.1)
            B = M // 450 if M % 450 == 0 else 1
            Lq = M // B
            Lp = (Lq + 127) // 128 * 128
            kw_r, kw_t = {}, {}
            if mode == "f32":
                out = torch.empty(M, N, device=DEV)
                kw = dict(mode=L.EPI_STORE_F32, bias=bias, out=out, ldc=N)
            elif mode == "t":
                out = torch.empty(M, N, device=DEV, dtype=BF)
                kw = dict(out=out, ldc=N)
            elif mode == "heads":
                imgs = [torch.zeros(B, 8, Lp, 64, device=DEV, dtype=BF) for _ in range(3)]
                kw = dict(mode=L.EPI_QKV_HEADS, out=imgs[0], out_k=imgs[1], out_v=imgs[2], scale_q=0.125, Lseq=Lq, Lp=Lp, H=8, n_q=512,
                          n_k=512, A2=A2, split_n=1024)
            elif mode == "act":
                out, out2 = torch.empty(M, N, device=DEV, dtype=BF), torch.empty(M, N, device=DEV, dtype=BF)
                kw = dict(bias=bias, out=out, ldc=N, out2=out2, ldc2=N, act2=L.ACT_GELU, seed=seed, site=22, thr=thr, drop_scale=sc)
            else:
                out, src = torch.empty(M, N, device=DEV, dtype=BF), torch.randn(M, N, device=DEV).to(BF)
                kw = dict(out=out, ldc=N, act_src=src, ld_src=N, act2=L.ACT_GELU, seed=seed, site=22, thr=thr, drop_scale=sc)
            t_tile = timeit(lambda: K.gemm_tile(DT, A, Wb, M, N, Kd, **kw))
            line = f"  {name:34s} N={N:5d} K={Kd:5d}: gemm_tile {t_tile:7.1f} us"
            fl = 2.0 * M * N * Kd
            for mt in (0, 4, 2, 1):
                if mt and (M + 16 * mt - 1) // (16 * mt) > 4096:
                    continue
                t = timeit(lambda: K.gemm_rows(A, ws, M, N, Kd, mt=mt, **kw))
                line += f" | rows mt={mt}: {t:7.1f} us ({fl / t / 1e6:6.0f} TF/s)"
            print(line, flush=True)
    # the packer: a decoder layer's seven linears, forward and input-gradient order
    ents, keep = [], []
    for _ in range(8):
        for (n, k) in ((1536, 512), (512, 512), (512, 512), (512, 512), (1024, 512), (512, 1024), (512, 512)):
            W = torch.randn(n, k, device=DEV)
            d1 = torch.empty(8, (n // 512) * (k // 32), 2048, device=DEV, dtype=BF)
            d2 = torch.empty(8, (k // 512) * (n // 32), 2048, device=DEV, dtype=BF)
            ents.append(dict(src=W, sn=k, sk=1, N=n, K=k, dst=d1))
            ents.append(dict(src=W, sn=1, sk=k, N=k, K=n, dst=d2))
            keep += [W, d1, d2]
    tab = K.ws_table(ents, DEV)
    print(f"pack_row_streams, 8 layers x 7 linears x 2 orders ({len(ents)} matrices): {timeit(lambda: K.pack_row_streams(tab)):.1f} us")


if __name__ == "__main__":
    main()
