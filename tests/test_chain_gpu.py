"""Row-block chain kernels (csrc/chain.hip, bf16) against the op-by-op kernels they replace (gemm_rowln / gemm_tile, which
are held to the reference by tests/test_kernels_gpu.py and tests/test_parity_gpu.py): same MFMA k order, so the GEMM
accumulators agree bit for bit and only LayerNorm summation order differs (then a bf16 rounding may flip)."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import tcdiff_oracle as O  # noqa: E402  (synthetic inputs only)
from tcdiff_amd import _lib as L  # noqa: E402
from tcdiff_amd import kernels as K  # noqa: E402
from tcdiff_amd.engine import DenoiserEngine  # noqa: E402
from tcdiff_amd.model import DanceDecoder  # noqa: E402

DEV = "cuda"


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def model_with(chain: int, dn, S):
    """chain: 0 = op-by-op kernels, 1 = chain A / cross-attention / chain B, 2 = the whole layer tail in one launch"""
    os.environ["TCDIFF_CHAIN"] = str(int(chain))
    try:
        m = DanceDecoder(nfeats=151, seq_len=S, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                         cond_feature_dim=438, activation=F.gelu, required_dancer_num=dn, compute_dtype="bf16")
        m.load_state_dict(O.synth_state_dict(dn=dn, seq_len=S))
        m.to(DEV).eval()
        m.engine(1)                       # the engine reads the switch when it is built
        assert m._engines[0].use_chain == (chain > 0) and m._engines[0].use_full == (chain == 2)
        return m
    finally:
        os.environ.pop("TCDIFF_CHAIN", None)


def folded(film, specs):
    """the FiLM rows as the chain kernels take them: every 1024-float block at offset `off` folded with the LayerNorm weights
    (g, b) in front of it -- or (None, linear2 bias) for the feed-forward block (kernels.fold_film)"""
    out = film.clone()
    for off, (g, b) in specs.items():
        out[:, off:off + 1024] = K.fold_film(film[:, off:off + 1024], g, b)
    return out


@pytest.mark.parametrize("dn,S,B", [(2, 60, 1), (2, 60, 3), (3, 150, 2), (3, 150, 5)])
def test_chained_network_matches_op_by_op_network(dn, S, B):
    """whole denoiser, both CFG branches (layer-0 rows shared between the branches, ragged last row block)"""
    Lq = dn * S
    cond = torch.stack([O.synth_cond(c, S) for c in range(B)]).to(DEV)
    x = torch.stack([O.synth_xT(c, Lq) for c in range(B)]).to(DEV)
    outs = []
    for chain in (0, 1, 2):
        m = model_with(chain, dn, S)
        tt = torch.full((B,), 640, dtype=torch.long, device=DEV)
        g = m.guided_forward(x, cond, tt, 2.0)
        f = m(x, cond, torch.arange(B, device=DEV) * 37 + 5, cond_drop_prob=0.0)
        outs.append((g, f))
    for which, other in (("chain", outs[1]), ("full chain", outs[2])):
      for a, b in zip(outs[0], other):
        d, mean = float((a - b).abs().max()), float((a - b).abs().mean())
        print(f"{which} vs op-by-op ({dn}x{S}, B={B}): max-abs {d:.2e}, mean-abs {mean:.2e}, |out| max {float(a.abs().max()):.2f}")
        assert torch.isfinite(b).all()
        # the two paths agree per kernel to a few flipped bf16 roundings (tests below).  A flipped element of a GEMM
        # operand moves the whole output row by ~1e-4, which flips ~2 % of the next roundings: within two layers the
        # two paths' rounding errors are independent, and their distance settles at ~sqrt(2) x the bf16-vs-fp32
        # deviation of either (max ~2e-2, mean ~2e-3: tests/test_parity_gpu.py), not beyond it.  Measured on MI355X:
        # max 1.1e-2, mean 2.1e-3.
        assert d < 3e-2 and mean < 4e-3


MTS = pytest.mark.parametrize("mt", [4, 2, 1])      # 16-row tiles per row block (tcdiff_chain_args.mt: 64 / 32 / 16-row blocks)
# waves per workgroup of the production launches (tcdiff_chain_args.nw): eight waves x 64 columns or four x 128 columns
NWS = pytest.mark.parametrize("nw", [8, 4])


@MTS
def test_chain_a_kernel_against_rowln_plus_tile(mt):
    """chain A alone on random data: x (fp32) and the Q image, incl. a_mod / xres_mod and a ragged tail block"""
    dt = L.DT_BF16
    Lq, nseq, H = 120, 3, 8
    M, Rs = nseq * Lq, 2 * Lq            # rows m and m % Rs share the input / residual rows for m >= Rs? (a_mod semantics)
    Lp = K.round_up(Lq, 128)
    Oa = rnd(Rs, 512, seed=1, scale=0.5).to(torch.bfloat16)
    Wfc = rnd(512, 512, seed=2, scale=512 ** -0.5).to(torch.bfloat16)
    Wq = rnd(512, 512, seed=3, scale=512 ** -0.5).to(torch.bfloat16)
    g1, b1 = 1 + 0.1 * rnd(512, seed=4), 0.1 * rnd(512, seed=5)
    g2, b2 = 1 + 0.1 * rnd(512, seed=6), 0.1 * rnd(512, seed=7)
    film = 0.3 * rnd(nseq, 2048, seed=8)
    xres = rnd(Rs, 512, seed=9)
    freqs = 1.0 / (10000 ** (torch.arange(0, 512, 2).float() / 512))
    rope = torch.empty(Lq, 512, device=DEV)
    K.rope_table(freqs.to(DEV), rope, Lq)
    # op-by-op
    x1 = torch.zeros(M, 512, device=DEV)
    rot = torch.zeros(M, 512, device=DEV, dtype=torch.bfloat16)
    K.gemm_rowln(dt, Oa, Wfc, M, 512, a_mod=Rs, flags=L.ROW_LN_POST | L.ROW_FILM | L.ROW_STORE_X | L.ROW_NEXT_LN | L.ROW_STORE_ROT,
                 ln_g=g1, ln_b=b1, ln_eps=1e-6, film=film, film_ld=2048, xres=xres, xres_mod=Rs, xout=x1, Lseq=Lq,
                 nln_g=g2, nln_b=b2, nln_eps=1e-5, rout=rot, rope=rope)
    Q1 = torch.zeros(nseq, H, Lp, 64, device=DEV, dtype=torch.bfloat16)
    K.gemm_tile(dt, rot, Wq, M, 512, 512, mode=L.EPI_QKV_HEADS, out=Q1, scale_q=0.125, Lseq=Lq, Lp=Lp, H=H, n_q=512, n_k=0)
    # chain
    ws = torch.cat([DenoiserEngine._stages_n512(Wfc), DenoiserEngine._stages_n512(Wq)], 1).contiguous()
    x2 = torch.zeros(M, 512, device=DEV)
    Q2 = torch.zeros_like(Q1)
    # the chain kernels take the rotary table and keep the residual stream COLUMN-BLOCKED ([64][rows][8]); layer 0's
    # residual input (gemm_rowln's output) is row-major
    K.chain(L.CHAIN_A, M, Lq, Oa, ws, mt=mt, a_mod=Rs, ln_eps=1e-6, film=folded(film, {0: (g1, b1)}), film_ld=2048, xres=xres,
            xres_mod=Rs, xres_rowmajor=True, xout=x2, n2_g=g2, n2_b=b2, n2_eps=1e-5, rope=K.to_cb(rope), q_out=Q2,
            scale_q=0.125, Lp=Lp, H=H)
    torch.cuda.synchronize()
    x2 = K.from_cb(x2, M)
    dx = float((x1 - x2).abs().max())
    dq = float((Q1.float() - Q2.float()).abs().max())
    print(f"chain A: x max-abs diff {dx:.2e} (|x| max {float(x1.abs().max()):.2f}), Q max-abs diff {dq:.2e}")
    assert dx < 1e-5
    assert dq < 2e-2 and float((Q1.float() - Q2.float()).abs().mean()) < 1e-4     # isolated bf16 rounding flips only
    assert float(Q2[:, :, Lq:].abs().max()) == 0.0                                   # padding rows untouched


@MTS
@pytest.mark.parametrize("last", [False, True])
def test_chain_b_kernel_against_op_by_op_sequence(last, mt):
    """chain B alone on random data against gemm_rowln / gemm_tile launched in the engine's op-by-op order"""
    dt = L.DT_BF16
    Lq, nseq, H = 120, 3, 8
    M = nseq * Lq - 7                     # ragged: the last sequence is short and the last block has a tail
    Lp = K.round_up(Lq, 128)
    bf = torch.bfloat16
    Oa = rnd(M, 512, seed=11, scale=0.5).to(bf)
    W = {n: rnd(*s, seed=20 + i, scale=s[1] ** -0.5).to(bf) for i, (n, s) in enumerate(
        [("cfc", (512, 512)), ("ff1", (1024, 512)), ("ff2", (512, 1024)), ("l3", (512, 512)), ("qkv", (1536, 512))])}
    vec = lambda seed, base=0.0, amp=0.1: base + amp * rnd(512, seed=seed)
    g1, b1, g3, b3n, g4, b4, gn, bn = vec(30, 1), vec(31), vec(32, 1), vec(33), vec(34, 1), vec(35), vec(36, 1), vec(37)
    bias1, bias2, bias3 = 0.05 * rnd(1024, seed=38), vec(39), vec(40)
    film = 0.3 * rnd(nseq, 4096, seed=41)
    xres = rnd(M, 512, seed=42)
    freqs = 1.0 / (10000 ** (torch.arange(0, 512, 2).float() / 512))
    rope = torch.empty(Lq, 512, device=DEV)
    K.rope_table(freqs.to(DEV), rope, Lq)
    z = lambda *s, dtype=bf: torch.zeros(*s, device=DEV, dtype=dtype)
    # ---- op-by-op (engine.network order)
    xa = xres.clone()
    h, h1, rot = z(M, 512), z(M, 1024), z(M, 512)
    K.gemm_rowln(dt, Oa, W["cfc"], M, 512, flags=L.ROW_LN_POST | L.ROW_FILM | L.ROW_STORE_X | L.ROW_NEXT_LN | L.ROW_STORE_H,
                 ln_g=g1, ln_b=b1, ln_eps=1e-6, film=film, film_ld=4096, xres=xa, xout=xa, Lseq=Lq, nln_g=g3, nln_b=b3n,
                 nln_eps=1e-5, hout=h)
    K.gemm_tile(dt, h, W["ff1"], M, 1024, 512, bias=bias1, act=L.ACT_GELU, out=h1, ldc=1024)
    K.gemm_rowln(dt, h1, W["ff2"], M, 1024, flags=L.ROW_BIAS | L.ROW_FILM | L.ROW_NEXT_LN | L.ROW_STORE_H, bias=bias2,
                 film=film[:, 2048:], film_ld=4096, xres=xa, Lseq=Lq, nln_g=g4, nln_b=b4, nln_eps=1e-5, hout=h)
    Q1, K1, V1 = z(nseq, H, Lp, 64), z(nseq, H, Lp, 64), z(nseq, H, Lp, 64)
    hl1 = z(M, 512)
    if last:
        K.gemm_rowln(dt, h, W["l3"], M, 512, bias=bias3, Lseq=Lq, flags=L.ROW_BIAS | L.ROW_STORE_H, hout=hl1)
    else:
        K.gemm_rowln(dt, h, W["l3"], M, 512, bias=bias3, xout=xa, Lseq=Lq,
                     flags=L.ROW_BIAS | L.ROW_STORE_X | L.ROW_NEXT_LN | L.ROW_STORE_H | L.ROW_STORE_ROT, nln_g=gn,
                     nln_b=bn, nln_eps=1e-5, hout=h, rout=rot, rope=rope)
        K.gemm_tile(dt, rot, W["qkv"], M, 1536, 512, A2=h, split_n=1024, mode=L.EPI_QKV_HEADS, out=Q1, out_k=K1,
                    out_v=V1, scale_q=0.125, Lseq=Lq, Lp=Lp, H=H, n_q=512, n_k=512)
    # ---- chain
    E = DenoiserEngine
    f1, f2 = E._stages_ff1(W["ff1"]), E._stages_ff2(W["ff2"])
    parts = [E._stages_n512(W["cfc"])]
    parts += E._ffn_order(f1, f2)
    parts.append(E._stages_n512(W["l3"]))
    if not last:
        parts += [E._stages_n512(W["qkv"][i * 512:(i + 1) * 512]) for i in range(3)]
    ws = torch.cat(parts, 1).contiguous()
    xb = K.to_cb(xres)                     # column-blocked residual stream, updated in place
    Q2, K2, V2, hl2 = z(nseq, H, Lp, 64), z(nseq, H, Lp, 64), z(nseq, H, Lp, 64), z(M, 512)
    ff = folded(film, {0: (g1, b1), 2048: (None, bias2)})
    K.chain(L.CHAIN_B_LAST if last else L.CHAIN_B, M, Lq, Oa, ws, mt=mt, ln_eps=1e-6, film=ff,
            film_ld=4096, xres=xb, xout=xb, n2_g=g3, n2_b=b3n, n2_eps=1e-5, rope=K.to_cb(rope), b1=bias1,
            film3=ff[:, 2048:], n4_g=g4, n4_b=b4, n4_eps=1e-5, b3=bias3, nn_g=None if last else gn,
            nn_b=None if last else bn, nn_eps=1e-5, q_out=None if last else Q2, k_out=None if last else K2,
            v_out=None if last else V2, h_out=hl2 if last else None, scale_q=0.125, Lp=Lp, H=H)
    torch.cuda.synchronize()
    md = lambda a, b: (float((a.float() - b.float()).abs().max()), float((a.float() - b.float()).abs().mean()))
    if last:
        d = md(hl1, hl2)
        print(f"chain B (last): linear3 rows max/mean diff {d[0]:.2e}/{d[1]:.2e} (|h| max {float(hl1.float().abs().max()):.2f})")
        assert d[0] < 6e-2 and d[1] < 2e-3
    else:
        dx = md(xa, K.from_cb(xb, M))
        print(f"chain B: x' max/mean diff {dx[0]:.2e}/{dx[1]:.2e} (|x| max {float(xa.abs().max()):.2f})")
        for nm, a, b in (("Q", Q1, Q2), ("K", K1, K2), ("V", V1, V2)):
            d = md(a, b)
            print(f"chain B: {nm} image max/mean diff {d[0]:.2e}/{d[1]:.2e} (max |.| {float(a.float().abs().max()):.2f})")
            assert d[0] < 1e-1 and d[1] < 3e-3
        assert dx[0] < 6e-2 and dx[1] < 2e-3


@pytest.mark.parametrize("S", [60, 150])      # 62 keys: the rolled cross-attention loop; 152 keys = 5 tiles: the pipelined one
@MTS
@NWS
@pytest.mark.parametrize("last", [False, True])
def test_fused_layer_chain_against_chain_a_attention_chain_b(last, S, mt, nw):
    """TC_CHAIN_FULL (cross-attention inside the launch, K / V from the fragment-ordered images) against the three
    launches it replaces on the same random data: blocks that straddle two sequences (L = 120: the second 32-row tile of
    block 1 crosses a sequence boundary), a shared null-conditioning slot for the first sequences, a ragged tail."""
    dt, bf = L.DT_BF16, torch.bfloat16
    Lq, nseq, H = 120, 5, 8
    M = nseq * Lq - 9
    Lp, Lk = K.round_up(Lq, 128), S + 2
    Lpc, nkt = K.round_up(Lk, 128), (Lk + 31) // 32
    n_shared, n_kv = 2, nseq - 2 + 1
    Oa = rnd(M, 512, seed=51, scale=0.5).to(bf)
    W = {n: rnd(*s, seed=60 + i, scale=s[1] ** -0.5).to(bf) for i, (n, s) in enumerate(
        [("sfc", (512, 512)), ("cq", (512, 512)), ("cfc", (512, 512)), ("ff1", (1024, 512)), ("ff2", (512, 1024)),
         ("l3", (512, 512)), ("qkv", (1536, 512))])}
    vec = lambda seed, base=0.0, amp=0.1: base + amp * rnd(512, seed=seed)
    gs = [vec(70 + i, 1 if i % 2 == 0 else 0) for i in range(12)]
    bias1, bias2, bias3 = 0.05 * rnd(1024, seed=90), vec(91), vec(92)
    film = 0.3 * rnd(nseq, 6144, seed=93)
    xres = rnd(M, 512, seed=94)
    rope = torch.empty(Lq, 512, device=DEV)
    K.rope_table((1.0 / (10000 ** (torch.arange(0, 512, 2).float() / 512))).to(DEV), rope, Lq)
    rope = K.to_cb(rope)                   # column-blocked, like the residual stream x1 / x2 below
    Kc = torch.zeros(n_kv, H, Lpc, 64, device=DEV, dtype=bf)
    Vc = torch.zeros_like(Kc)
    Kc[:, :, :Lk] = rnd(n_kv, H, Lk, 64, seed=95).to(bf)
    Vc[:, :, :Lk] = rnd(n_kv, H, Lk, 64, seed=96).to(bf)
    z = lambda *s, dtype=bf: torch.zeros(*s, device=DEV, dtype=dtype)
    E = DenoiserEngine
    f1, f2 = E._stages_ff1(W["ff1"]), E._stages_ff2(W["ff2"])
    partsB = [E._stages_n512(W["cfc"])]
    partsB += E._ffn_order(f1, f2)
    partsB.append(E._stages_n512(W["l3"]))
    if not last:
        partsB += [E._stages_n512(W["qkv"][i * 512:(i + 1) * 512]) for i in range(3)]
    wsA = torch.cat([E._stages_n512(W["sfc"]), E._stages_n512(W["cq"])], 1).contiguous()
    wsB = torch.cat(partsB, 1).contiguous()
    g1, g2 = E._stages_ff1(W["ff1"], nw), E._stages_ff2(W["ff2"], nw)          # the fused launch's stream, in its wave form
    partsF = [E._stages_n512(W["sfc"], nw), E._stages_n512(W["cq"], nw), E._stages_n512(W["cfc"], nw)] + E._ffn_order(g1, g2)
    partsF.append(E._stages_n512(W["l3"], nw))
    if not last:
        partsF += [E._stages_n512(W["qkv"][i * 512:(i + 1) * 512], nw) for i in range(3)]
    wsF = torch.cat(partsF, 1).contiguous()
    assert wsF.shape == (nw, wsA.shape[1] + wsB.shape[1], 16384 // nw)
    ff = folded(film, {0: (gs[0], gs[1]), 2048: (gs[4], gs[5]), 4096: (None, bias2)})
    tail = dict(b1=bias1, film3=ff[:, 4096:], n4_g=gs[6], n4_b=gs[7], b3=bias3, nn_g=None if last else gs[8],
                nn_b=None if last else gs[9], scale_q=0.125, Lp=Lp, H=H)
    # ---- three launches
    x1 = K.to_cb(xres)
    Qc, O2 = z(nseq, H, Lp, 64), z(M, 512)
    K.chain(L.CHAIN_A, M, Lq, Oa, wsA, mt=mt, ln_eps=1e-6, film=ff, film_ld=6144, xres=x1, xout=x1,
            n2_g=gs[2], n2_b=gs[3], rope=rope, q_out=Qc, scale_q=0.125, Lp=Lp, H=H)
    K.attention(dt, Qc, Kc, Vc, O2, nseq, H, Lq, Lk, Lp, Lpc, 512, n_shared=n_shared)
    Q1, K1, V1, h1 = z(nseq, H, Lp, 64), z(nseq, H, Lp, 64), z(nseq, H, Lp, 64), z(M, 512)
    K.chain(L.CHAIN_B_LAST if last else L.CHAIN_B, M, Lq, O2, wsB, mt=mt, ln_eps=1e-6,
            film=ff[:, 2048:], film_ld=6144, xres=x1, xout=x1, n2_g=gs[10], n2_b=gs[11], rope=rope,
            q_out=None if last else Q1, k_out=None if last else K1, v_out=None if last else V1,
            h_out=h1 if last else None, **tail)
    # ---- one launch
    Kf, Vf = z(n_kv, H, nkt * 2048), z(n_kv, H, nkt * 2048)
    K.pack_kv_frags(Kc, Vc, Kf, Vf, n_kv, H, Lpc, nkt, 0, Lk)
    x2 = K.to_cb(xres)
    Q2, K2, V2, h2 = z(nseq, H, Lp, 64), z(nseq, H, Lp, 64), z(nseq, H, Lp, 64), z(M, 512)
    K.chain(L.CHAIN_FULL_LAST if last else L.CHAIN_FULL, M, Lq, Oa, wsF, mt=mt, ln_eps=1e-6,
            film=ff, film_ld=6144, xres=x2, xout=x2, n2_g=gs[2], n2_b=gs[3], rope=rope,
            filmb=ff[:, 2048:], n3_g=gs[10], n3_b=gs[11], kf=Kf, vf=Vf, n_shared=n_shared, nkt=nkt, Lk=Lk,
            q_out=None if last else Q2, k_out=None if last else K2, v_out=None if last else V2,
            h_out=h2 if last else None, **tail)
    torch.cuda.synchronize()
    md = lambda a, b: (float((a.float() - b.float()).abs().max()), float((a.float() - b.float()).abs().mean()))
    pairs = [("h", h1, h2)] if last else [("x'", x1, x2), ("Q", Q1, Q2), ("K", K1, K2), ("V", V1, V2)]
    for nm, a, b in pairs:
        d = md(a, b)
        print(f"fused layer chain ({'last' if last else 'mid'}): {nm} max/mean diff {d[0]:.2e}/{d[1]:.2e} (max |.| {float(a.float().abs().max()):.2f})")
        # the in-kernel softmax sums in another order and feeds O straight on: isolated bf16 flips, spread by the GEMMs
        assert d[0] < 1.5e-1 and d[1] < 4e-3


@MTS
@NWS
def test_front_chain_against_rowln_plus_qkv_tile(mt, nw):
    """TC_CHAIN_FRONT (last fusion linear of each dancer over 64-frame blocks + layer 0's norm1 / rotary / Q, K, V) against
    the two launches it replaces, gemm_rowln with dancer groups and the QKV gemm_tile, on random data: 3 dancers, frames
    per sequence not a multiple of anything (S = 70: blocks straddle sequences), a ragged last block."""
    dt, bf = L.DT_BF16, torch.bfloat16
    dn, S, nseq, H = 3, 70, 3, 8
    Lq, Mf = dn * S, nseq * S                    # tokens per sequence, frames
    Rs = Mf * dn
    Lp = K.round_up(Lq, 128)
    f2 = rnd(Mf, 1024, seed=101, scale=0.5).to(bf)
    W3 = rnd(512 * dn, 1024, seed=102, scale=1024 ** -0.5).to(bf)
    b3 = 0.1 * rnd(512 * dn, seed=103)
    Wqkv = rnd(1536, 512, seed=104, scale=512 ** -0.5).to(bf)
    g1, b1 = 1 + 0.1 * rnd(512, seed=105), 0.1 * rnd(512, seed=106)
    rope = torch.empty(Lq, 512, device=DEV)
    K.rope_table((1.0 / (10000 ** (torch.arange(0, 512, 2).float() / 512))).to(DEV), rope, Lq)
    z = lambda *s_, dtype=bf: torch.zeros(*s_, device=DEV, dtype=dtype)
    # ---- the two launches
    xs1, h, rot = z(Rs, 512, dtype=torch.float32), z(Rs, 512), z(Rs, 512)
    K.gemm_rowln(dt, f2, W3, Mf, 1024, bias=b3, xout=xs1, Lseq=Lq,
                 flags=L.ROW_BIAS | L.ROW_STORE_X | L.ROW_NEXT_LN | L.ROW_STORE_H | L.ROW_STORE_ROT, nln_g=g1, nln_b=b1,
                 nln_eps=1e-5, hout=h, rout=rot, rope=rope, out_mul=dn, out_add=0, groups=dn)
    Q1, K1, V1 = z(nseq, H, Lp, 64), z(nseq, H, Lp, 64), z(nseq, H, Lp, 64)
    K.gemm_tile(dt, rot, Wqkv, Rs, 1536, 512, A2=h, split_n=1024, mode=L.EPI_QKV_HEADS, out=Q1, out_k=K1, out_v=V1,
                scale_q=0.125, Lseq=Lq, Lp=Lp, H=H, n_q=512, n_k=512)
    # ---- one launch
    E = DenoiserEngine
    tail = [E._stages_n512(Wqkv[i * 512:(i + 1) * 512], nw) for i in range(3)]
    ws = torch.stack([torch.cat([E._stages_n512(W3[512 * d:512 * d + 512], nw)] + tail, 1) for d in range(dn)]).contiguous()
    assert ws.shape == (dn, nw, 80, 16384 // nw)
    xs2 = z(Rs, 512, dtype=torch.float32)
    Q2, K2, V2 = z(nseq, H, Lp, 64), z(nseq, H, Lp, 64), z(nseq, H, Lp, 64)
    K.chain(L.CHAIN_FRONT, Mf, Lq, f2, ws, mt=mt, b3=b3, nn_g=g1, nn_b=b1, nn_eps=1e-5, rope=K.to_cb(rope), xout=xs2, q_out=Q2,
            k_out=K2, v_out=V2, scale_q=0.125, Lp=Lp, H=H, dn=dn)
    torch.cuda.synchronize()
    md = lambda a_, b_: (float((a_.float() - b_.float()).abs().max()), float((a_.float() - b_.float()).abs().mean()))
    dx = md(xs1, K.from_cb(xs2, Rs))
    print(f"front chain: x max/mean diff {dx[0]:.2e}/{dx[1]:.2e} (|x| max {float(xs1.abs().max()):.2f})")
    assert dx[0] < 1e-4                                          # fp32 accumulators of the same bf16 products
    for nm, a_, b_ in (("Q", Q1, Q2), ("K", K1, K2), ("V", V1, V2)):
        d = md(a_, b_)
        print(f"front chain: {nm} image max/mean diff {d[0]:.2e}/{d[1]:.2e} (max |.| {float(a_.float().abs().max()):.2f})")
        assert d[0] < 6e-2 and d[1] < 5e-4                          # isolated bf16 rounding flips
        assert float(b_[:, :, Lq:].abs().max()) == 0.0              # padding rows untouched


@MTS
@NWS
def test_fused_layer_chain_against_a_torch_evaluation_of_the_layer_tail(mt, nw):
    """ONE TC_CHAIN_FULL launch against a plain torch (float64) evaluation of model/model.py:103-106,327,331-344 and the next
    layer's :326,374-383 on the same bf16 operands -- not against another kernel of this library.  The reference rounds to
    bf16 exactly where the kernel hands an activation to an MFMA (GEMM / attention operands), nowhere else."""
    bf, D = torch.bfloat16, torch.float64
    Lq, nseq, H, S = 120, 5, 8, 60
    M = nseq * Lq - 9
    Lp, Lk = K.round_up(Lq, 128), S + 2
    Lpc, nkt = K.round_up(Lk, 128), (Lk + 31) // 32
    n_shared, n_kv = 2, nseq - 2 + 1
    Oa = rnd(M, 512, seed=151, scale=0.5).to(bf)
    W = {n: rnd(*s, seed=160 + i, scale=s[1] ** -0.5).to(bf) for i, (n, s) in enumerate(
        [("sfc", (512, 512)), ("cq", (512, 512)), ("cfc", (512, 512)), ("ff1", (1024, 512)), ("ff2", (512, 1024)),
         ("l3", (512, 512)), ("qkv", (1536, 512))])}
    vec = lambda seed, base=0.0, amp=0.1: base + amp * rnd(512, seed=seed)
    gs = [vec(170 + i, 1 if i % 2 == 0 else 0) for i in range(12)]
    bias1, bias2, bias3 = 0.05 * rnd(1024, seed=190), vec(191), vec(192)
    film = 0.3 * rnd(nseq, 6144, seed=193)
    xres = rnd(M, 512, seed=194)
    rope_rm = torch.empty(Lq, 512, device=DEV)
    K.rope_table((1.0 / (10000 ** (torch.arange(0, 512, 2).float() / 512))).to(DEV), rope_rm, Lq)
    Kc = torch.zeros(n_kv, H, Lpc, 64, device=DEV, dtype=bf)
    Vc = torch.zeros_like(Kc)
    Kc[:, :, :Lk] = rnd(n_kv, H, Lk, 64, seed=195).to(bf)
    Vc[:, :, :Lk] = rnd(n_kv, H, Lk, 64, seed=196).to(bf)
    z = lambda *s, dtype=bf: torch.zeros(*s, device=DEV, dtype=dtype)
    E = DenoiserEngine
    f1, f2 = E._stages_ff1(W["ff1"], nw), E._stages_ff2(W["ff2"], nw)
    parts = [E._stages_n512(W["sfc"], nw), E._stages_n512(W["cq"], nw), E._stages_n512(W["cfc"], nw)]
    parts += E._ffn_order(f1, f2)
    parts.append(E._stages_n512(W["l3"], nw))
    parts += [E._stages_n512(W["qkv"][i * 512:(i + 1) * 512], nw) for i in range(3)]
    wsF = torch.cat(parts, 1).contiguous()
    Kf, Vf = z(n_kv, H, nkt * 2048), z(n_kv, H, nkt * 2048)
    K.pack_kv_frags(Kc, Vc, Kf, Vf, n_kv, H, Lpc, nkt, 0, Lk)
    x2 = K.to_cb(xres)
    Q2, K2, V2 = z(nseq, H, Lp, 64), z(nseq, H, Lp, 64), z(nseq, H, Lp, 64)
    ff = folded(film, {0: (gs[0], gs[1]), 2048: (gs[4], gs[5]), 4096: (None, bias2)})
    K.chain(L.CHAIN_FULL, M, Lq, Oa, wsF, mt=mt, ln_eps=1e-6, film=ff, film_ld=6144, xres=x2,
            xout=x2, n2_g=gs[2], n2_b=gs[3], rope=K.to_cb(rope_rm), filmb=ff[:, 2048:], n3_g=gs[10],
            n3_b=gs[11], kf=Kf, vf=Vf, n_shared=n_shared, nkt=nkt, Lk=Lk, q_out=Q2, k_out=K2, v_out=V2, b1=bias1,
            film3=ff[:, 4096:], n4_g=gs[6], n4_b=gs[7], b3=bias3, nn_g=gs[8], nn_b=gs[9], scale_q=0.125, Lp=Lp, H=H)
    torch.cuda.synchronize()
    # ---- torch, float64, on the CPU ---------------------------------------------------------------------------------------
    c = lambda t: t.detach().cpu().to(D)
    rb = lambda t: t.to(bf).to(D)                        # the kernel's bf16 hand-offs
    ln = lambda v, g, b, eps: F.layer_norm(v, (512,), c(g), c(b), eps)
    seq = torch.arange(M) // Lq
    pos = torch.arange(M) % Lq
    cs = c(rope_rm)[pos].reshape(M, 256, 2)

    def rot(u):
        up = u.reshape(M, 256, 2)
        return torch.stack((up[..., 0] * cs[..., 0] - up[..., 1] * cs[..., 1], up[..., 1] * cs[..., 0] + up[..., 0] * cs[..., 1]), -1).reshape(M, 512)

    def aff(y, blk):
        fm = c(film)[seq]
        return (fm[:, 2048 * blk:2048 * blk + 512] + 1) * y + fm[:, 2048 * blk + 512:2048 * blk + 1024]
    Wd = {k: c(v) for k, v in W.items()}
    x = c(xres) + aff(ln(c(Oa) @ Wd["sfc"].t(), gs[0], gs[1], 1e-6), 0)
    r2 = rb(rot(ln(x, gs[2], gs[3], 1e-5)))
    q = rb((r2 @ Wd["cq"].t()) * 0.125).reshape(M, H, 64)
    kv = torch.where(seq < n_shared, torch.zeros_like(seq), seq - n_shared + 1)
    Kd, Vd = c(Kc)[:, :, :Lk], c(Vc)[:, :, :Lk]
    sc = torch.einsum("mhd,mhkd->mhk", q, Kd[kv])
    pu = torch.exp(sc - sc.amax(-1, keepdim=True))       # the kernel's P is the UNNORMALISED exp (a bf16 MFMA operand);
    oc = rb((torch.einsum("mhk,mhkd->mhd", rb(pu), Vd[kv]) / pu.sum(-1, keepdim=True)).reshape(M, 512))   # 1 / l comes last
    x = x + aff(ln(oc @ Wd["cfc"].t(), gs[4], gs[5], 1e-6), 1)
    h3 = rb(ln(x, gs[10], gs[11], 1e-5))
    a1 = rb(F.gelu(h3 @ Wd["ff1"].t() + c(bias1)))
    x = x + aff(a1 @ Wd["ff2"].t() + c(bias2), 2)
    h4 = rb(ln(x, gs[6], gs[7], 1e-5))
    xn = h4 @ Wd["l3"].t() + c(bias3)
    hn = ln(xn, gs[8], gs[9], 1e-5)
    rn, hb = rb(rot(hn)), rb(hn)
    Qr, Kr, Vr = (rn @ Wd["qkv"][:512].t()) * 0.125, rn @ Wd["qkv"][512:1024].t(), hb @ Wd["qkv"][1024:].t()
    got_x = K.from_cb(x2, M).cpu().to(D)
    dx = (got_x - xn).abs()
    print(f"fused layer chain vs torch float64: x' max-abs {float(dx.max()):.2e}, mean-abs {float(dx.mean()):.2e} (|x'| max {float(xn.abs().max()):.2f})")
    # what is left: bf16 hand-offs that land on the other side of a rounding boundary (one ulp = 2^-8 relative; the online
    # softmax rounds P against a running maximum, torch against the final one), each spread over a row by the next GEMM
    assert float(dx.max()) < 3e-2 and float(dx.mean()) < 4e-3
    for nm, img, ref in (("Q", Q2, Qr), ("K", K2, Kr), ("V", V2, Vr)):
        g = img.cpu().to(D)
        tok = torch.stack([g[s_, :, :min(Lq, M - s_ * Lq)].transpose(0, 1).reshape(-1, 512) for s_ in range(nseq)]).reshape(-1, 512) \
            if M == nseq * Lq else torch.cat([g[s_, :, :min(Lq, M - s_ * Lq)].transpose(0, 1).reshape(-1, 512) for s_ in range(nseq)])
        d = (tok - ref).abs()
        print(f"   {nm} image max-abs {float(d.max()):.2e}, mean-abs {float(d.mean()):.2e} (max |.| {float(ref.abs().max()):.2f})")
        assert float(d.max()) < 8e-2 and float(d.mean()) < 4e-3
