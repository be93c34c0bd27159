"""Diagnostic: per-phase time line of chain B (block 0, wave 0) from a -DCH_STAMP build (tools/chain_ablate.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TCDIFF_LIB_PATH", "tools/probe/libtc_STAMP.so")
import torch
exec(open(os.path.join(os.path.dirname(__file__), "chain_bench.py")).read().split("def run(")[0])
names = {0: "start", 1: "A block + first stages landed", 2: "fc GEMM", 3: "fc epilogue: stats + LN/FiLM/res/store", 4: "norm3 stats", 5: "norm3 -> LDS + barrier",
         22: "linear2 epilogue (loads)", 23: "norm4 stats", 24: "norm4 -> LDS + barrier", 25: "linear3 GEMM", 26: "linear3 epilogue", 27: "norm1' stats",
         28: "norm1' + rotary -> LDS", 29: "Q GEMM", 30: "Q store + K GEMM", 31: "K store", 32: "V GEMM", 33: "V store + drain"}
for c in range(4):
    names[6 + 4 * c] = f"linear1 chunk {c} GEMM"; names[7 + 4 * c] = "  barrier"; names[8 + 4 * c] = "  GELU -> LDS + barrier"; names[9 + 4 * c] = f"linear2 chunk {c} GEMM"
for nblk in (1, 225):
    M = nblk * 64
    nseq = (M + Lq - 1) // Lq
    Oa = rnd(M, 512, scale=0.5).to(bf); film = 0.3 * rnd(nseq, 4096); x = rnd(M, 512)
    Q, Kk, V = (torch.zeros(nseq, H, Lp, 64, device=dev, dtype=bf) for _ in range(3))
    st = torch.zeros(64, device=dev, dtype=torch.int64)
    for _ in range(5):
        K.chain(L.CHAIN_B, 288, M, Lq, Oa, wsB, ln_g=g[0], ln_b=g[1], ln_eps=1e-6, film=film, film_ld=4096, xres=x, xout=x,
                n2_g=g[2], n2_b=g[3], rope=rope, b1=b1, b2=b2, film3=film[:, 2048:], n4_g=g[4], n4_b=g[5], b3=b3, nn_g=g[6],
                nn_b=g[7], q_out=Q, k_out=Kk, v_out=V, h_out=st, Lp=Lp, H=H)
    torch.cuda.synchronize()
    t = st.cpu().tolist()
    print(f"---- chain B, {nblk} block(s): total {(t[33] - t[0]) / 100:.1f} us (100 MHz counter)")
    prev = t[0]
    for i in range(1, 34):
        print(f"  {names.get(i, str(i)):45s} {(t[i] - prev) / 100:6.2f} us")
        prev = t[i]
