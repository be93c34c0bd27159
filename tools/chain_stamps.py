"""Diagnostic: per-phase time line of the fused layer chain (block 0, wave 0) from a -DCH_STAMP build."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("TCDIFF_LIB_PATH", "tools/probe/libtc_STAMP.so")
import torch
exec(open(os.path.join(os.path.dirname(__file__), "chain_bench.py")).read().split("def run(")[0])
S_, nkt = 150, 5
if os.environ.get("XATT_KEYS"):      # diagnostic: the in-kernel attention over a longer key set (e.g. 450 = a self-attention's)
    S_ = int(os.environ["XATT_KEYS"]) - 2
    nkt = (S_ + 2 + 31) // 32
order = [(0, "start"), (1, "A block / self-attention + first stages landed"), (2, "self fc GEMM"), (40, "  (row loads issued)"), (41, "  LN statistics exchange (waits for the SIMD's second wave)"),
         (3, "  LN, FiLM, residual, store"),
         (4, "norm2 stats"), (34, "norm2 + rotary -> LDS + barrier"), (35, "w_qs GEMM"), (36, "cross-attention (incl. barrier)"),
         (37, "cross fc GEMM (incl. barrier)"), (42, "  (row loads issued)"), (43, "  LN statistics exchange"), (38, "  LN, FiLM, residual, store"), (39, "norm3 stats"), (5, "norm3 -> LDS, consts, 2 barriers")]
for c in range(4):
    order += [(6 + 4 * c, f"linear1 chunk {c} GEMM"), (8 + 4 * c, "  GELU -> LDS + barrier"), (9 + 4 * c, f"linear2 chunk {c} GEMM")]
order += [(22, "linear2 epilogue"), (23, "norm4 stats (incl. barrier)"), (24, "norm4 -> LDS + barrier"), (25, "linear3 GEMM"),
          (26, "linear3 epilogue (incl. barrier)"), (27, "norm1' stats"), (28, "norm1' + rotary -> LDS + barrier"), (29, "Q GEMM"),
          (30, "Q store + K GEMM"), (31, "K store"), (32, "V GEMM"), (33, "V store")]
NW = int(os.environ.get("NW", "8"))                  # wave form of the launch (tcdiff_chain_args.nw)
def _stream(nw):
    f1_, f2_ = E._stages_ff1(W["ff1"], nw), E._stages_ff2(W["ff2"], nw)
    pp = [E._stages_n512(W["sfc"], nw), E._stages_n512(W["cq"], nw), E._stages_n512(W["cfc"], nw)] + E._ffn_order(f1_, f2_)
    pp.append(E._stages_n512(W["l3"], nw))
    pp += [E._stages_n512(W["qkv"][i * 512:(i + 1) * 512], nw) for i in range(3)]
    return torch.cat(pp, 1).contiguous()
wsF = _stream(NW)
SA = os.environ.get("SA") == "1"    # the self-attention inside the launch: whole sequences, blocks cut per sequence (8 per 450 rows)
for nblk in ((8, 256) if SA else (1, 225)):
    M = nblk // 8 * Lq if SA else nblk * 64
    nseq = (M + Lq - 1) // Lq
    Oa = rnd(M, 512, scale=0.5).to(bf); film = 0.3 * rnd(nseq, 6144); x = rnd(M, 512)
    Q, Kk, V = (torch.zeros(nseq, H, Lp, 64, device=dev, dtype=bf) for _ in range(3))
    kf = rnd(nseq + 1, H, nkt * 2048, scale=0.5).to(bf); vf = rnd(nseq + 1, H, nkt * 2048, scale=0.5).to(bf)
    st = torch.zeros(8 * 64 + 8 * 32, device=dev, dtype=torch.int64)
    skt = (Lq + 31) // 32
    sa = {}
    if SA:
        qf = rnd(nblk, 8, 4, 2, 64, 8, scale=0.3).to(bf)
        skf, svf = rnd(2, nseq, H, skt * 2048, scale=0.5).to(bf), rnd(2, nseq, H, skt * 2048, scale=0.5).to(bf)
        sa = dict(seq_blocks=True, sa_q=qf, sa_kf=skf[0], sa_vf=svf[0], sa_nkt=skt, qf_out=qf, kf_out=skf[1], vf_out=svf[1], out_nkt=skt)
    for _ in range(5):
        K.chain(**sa, mode=L.CHAIN_FULL, M=M, Lseq=Lq, A=Oa, wstream=wsF, mt=4, ln_eps=1e-6, film=film, film_ld=6144, xres=x, xout=x,
                n2_g=g[2], n2_b=g[3], rope=rope, b1=b1, film3=film[:, 4096:], n4_g=g[4], n4_b=g[5], b3=b3, nn_g=g[6],
                nn_b=g[7], h_out=st, Lp=Lp, H=H, filmb=film[:, 2048:],
                n3_g=g[2], n3_b=g[3], kf=kf, vf=vf, n_shared=nseq // 2, nkt=nkt, Lk=S_ + 2, **({} if SA else dict(q_out=Q, k_out=Kk, v_out=V)))
    if False:
        K.chain(L.CHAIN_FULL, M, Lq, Oa, wsF, mt=4, ln_eps=1e-6, film=film, film_ld=6144, xres=x, xout=x,
                n2_g=g[2], n2_b=g[3], rope=rope, b1=b1, film3=film[:, 4096:], n4_g=g[4], n4_b=g[5], b3=b3, nn_g=g[6],
                nn_b=g[7], q_out=Q, k_out=Kk, v_out=V, h_out=st, Lp=Lp, H=H, filmb=film[:, 2048:],
                n3_g=g[2], n3_b=g[3], kf=kf, vf=vf, n_shared=nseq // 2, nkt=nkt, Lk=S_ + 2)
    torch.cuda.synchronize()
    tw = st.cpu()[:512].reshape(8, 64).tolist()[:NW]
    sg = st.cpu()[512:].reshape(8, 32).tolist()[:NW]
    t = tw[0]
    print(f"---- fused layer chain, {nblk} block(s): total {(t[33] - t[0]) / 100:.1f} us (100 MHz counter); per phase: wave 0's "
          f"duration, then every wave's arrival relative to wave 0 (us)")
    print(f"  last wave ends at {(max(tw[w][33] for w in range(NW)) - t[0]) / 100:.1f} us")
    print(f"  shader clock over the kernel (s_memtime / s_memrealtime): {(t[61] - t[60]) / max(1, t[33] - t[0]) * 100:.0f} MHz")
    prev = t[0]
    for i, name in order[1:]:
        rel = " ".join(f"{(tw[w][i] - t[i]) / 100:+5.2f}" for w in range(1, NW))
        print(f"  {name:50s} {(t[i] - prev) / 100:6.2f} us   [{rel}]")
        prev = t[i]
    print("  w_qs GEMM, shader cycles from the phase's start to the end of stage k, per wave (k = 0..15):")
    for w in range(NW):
        print(f"    wave {w}: " + " ".join(f"{sg[w][k] - sg[w][16]:6d}" for k in range(16)) + f"   (start: {sg[w][16] - tw[w][60]:6d} cycles after the kernel's)")
