"""Kernel-level parity (MI355X only): every launcher of the C ABI against an fp64/fp32 torch evaluation of the
same op on the same device.  f32 mode must match to fp32 rounding; bf16 mode is compared against a reference
fed the SAME bf16-rounded operands, so only accumulation order and the bf16 output rounding differ."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from tcdiff_amd import _lib as L  # noqa: E402
from tcdiff_amd import kernels as K  # noqa: E402

DEV = "cuda"
DTS = [L.DT_F32, L.DT_BF16]


def T(dt):
    return K.TORCH_DT[dt]


def tol(dt, f32=2e-4, bf16=2e-2):
    return f32 if dt == L.DT_F32 else bf16


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def relerr(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def act_ref(v, act):
    import torch.nn.functional as F
    return {L.ACT_NONE: lambda t: t, L.ACT_RELU: F.relu, L.ACT_GELU: F.gelu, L.ACT_MISH: F.mish, L.ACT_SILU: F.silu}[act](v)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,N,Kd,act", [(300, 200, 192, L.ACT_NONE), (128, 128, 64, L.ACT_GELU), (77, 1536, 512, L.ACT_RELU),
                                        (1000, 151, 512, L.ACT_MISH), (32, 640, 2048, L.ACT_SILU)])
def test_gemm_tile_store(dt, M, N, Kd, act):
    A = rnd(M, Kd, seed=1).to(T(dt))
    W = (rnd(N, Kd, seed=2) / math.sqrt(Kd)).to(T(dt))
    bias = rnd(N, seed=3)
    ref = act_ref(A.double() @ W.double().T + bias.double(), act)
    for mode, odt in ((L.EPI_STORE_F32, torch.float32), (L.EPI_STORE_T, T(dt))):
        ldc = N + 5
        out = torch.full((M, ldc), 7.0, device=DEV, dtype=odt)
        K.gemm_tile(dt, A, W, M, N, Kd, bias=bias, act=act, mode=mode, out=out, ldc=ldc)
        torch.cuda.synchronize()
        assert relerr(out[:, :N], ref) < tol(dt), (mode, relerr(out[:, :N], ref))
        assert bool((out[:, N:] == 7.0).all()), "wrote outside the N columns"


@pytest.mark.parametrize("M,N,Kd,act,a_mod", [(150, 1024, 576, L.ACT_RELU, 0), (150, 1536, 1024, L.ACT_NONE, 0), (60, 1024, 1024, L.ACT_RELU, 0),
                                              (7, 64, 64, L.ACT_GELU, 0), (33, 96, 2048, L.ACT_SILU, 0), (150, 512, 320, L.ACT_NONE, 50),
                                              (450, 1024, 1024, L.ACT_RELU, 0)])
def test_gemm_tile_small_products(M, N, Kd, act, a_mod):
    """The shapes tcdiff_gemm_tile hands to its small-M kernel when the caller allows it (tcdiff_tile_epi.small_m; csrc/gemm.hip
    gemm_small_kernel: bf16, plain epilogue, a 128 x 128 tiling that would leave the chip idle -- the sampler's input / fusion
    projections on a one-clip job, model/model.py:560-561; the last case is too large and stays on the 128 x 128 tiling):
    32 x 32 output tiles, K dealt to the eight waves, one exchange.  Row tails (M % 32), k-steps that do not divide by the waves
    (K = 576: 18 steps; K = 64: two waves work), a_mod, both output types, an output wider than N."""
    dt = L.DT_BF16
    A = rnd(a_mod if a_mod else M, Kd, seed=1).to(T(dt))
    W = (rnd(N, Kd, seed=2) / math.sqrt(Kd)).to(T(dt))
    bias = rnd(N, seed=3)
    rows = torch.arange(M, device=DEV) % a_mod if a_mod else torch.arange(M, device=DEV)
    ref = act_ref(A.double()[rows] @ W.double().T + bias.double(), act)
    for mode, odt in ((L.EPI_STORE_F32, torch.float32), (L.EPI_STORE_T, T(dt))):
        ldc = N + 4
        out = torch.full((M + 1, ldc), 7.0, device=DEV, dtype=odt)
        K.gemm_tile(dt, A, W, M, N, Kd, bias=bias, act=act, mode=mode, out=out, ldc=ldc, a_mod=a_mod, small_m=True)
        torch.cuda.synchronize()
        assert relerr(out[:M, :N], ref) < (1e-5 if mode == L.EPI_STORE_F32 else tol(dt)), (mode, relerr(out[:M, :N], ref))
        assert bool((out[:M, N:] == 7.0).all()) and bool((out[M] == 7.0).all()), "wrote outside the M x N block"


@pytest.mark.parametrize("dt", DTS)
def test_gemm_tile_split_and_amod(dt):
    M, Kd, N = 260, 128, 512
    A1, A2 = rnd(100, Kd, seed=4).to(T(dt)), rnd(100, Kd, seed=5).to(T(dt))
    W = (rnd(N, Kd, seed=6) / math.sqrt(Kd)).to(T(dt))
    out = torch.zeros(M, N, device=DEV)
    K.gemm_tile(dt, A1, W, M, N, Kd, A2=A2, split_n=256, a_mod=100, mode=L.EPI_STORE_F32, out=out, ldc=N)
    idx = torch.arange(M, device=DEV) % 100
    ref = torch.cat([A1.double()[idx] @ W.double()[:256].T, A2.double()[idx] @ W.double()[256:].T], 1)
    assert relerr(out, ref) < tol(dt)


@pytest.mark.parametrize("dt", DTS)
def test_gemm_tile_qkv_heads(dt):
    Lq, nseq, H, Lp = 70, 3, 8, 128
    M, Kd = nseq * Lq, 128
    A1, A2 = rnd(M, Kd, seed=7).to(T(dt)), rnd(M, Kd, seed=8).to(T(dt))
    W = (rnd(1536, Kd, seed=9) / math.sqrt(Kd)).to(T(dt))
    bias = rnd(1536, seed=10)
    Q = torch.zeros(nseq + 1, H, Lp, 64, device=DEV, dtype=T(dt))
    Kk = torch.zeros_like(Q)
    Vv = torch.zeros_like(Q)
    K.gemm_tile(dt, A1, W, M, 1536, Kd, A2=A2, split_n=1024, bias=bias, mode=L.EPI_QKV_HEADS, out=Q, out_k=Kk,
                out_v=Vv, scale_q=0.125, Lseq=Lq, Lp=Lp, H=H, n_q=512, n_k=512, tok_off=2, seq_off=1)
    q = ((A1.double() @ W.double()[:512].T + bias.double()[:512]) * 0.125).view(nseq, Lq, H, 64).permute(0, 2, 1, 3)
    k = (A1.double() @ W.double()[512:1024].T + bias.double()[512:1024]).view(nseq, Lq, H, 64).permute(0, 2, 1, 3)
    v = (A2.double() @ W.double()[1024:].T + bias.double()[1024:]).view(nseq, Lq, H, 64).permute(0, 2, 1, 3)
    assert relerr(Q[1:, :, 2:2 + Lq], q) < tol(dt)
    assert relerr(Kk[1:, :, 2:2 + Lq], k) < tol(dt)
    assert relerr(Vv[1:, :, 2:2 + Lq], v) < tol(dt)
    assert float(Q[0].abs().max()) == 0 and float(Vv[0].abs().max()) == 0
    assert float(Vv[1:, :, 2 + Lq:].abs().max()) == 0 and float(Q[1:, :, :2].abs().max()) == 0


# ---- large problems (thousands of tiles, ragged edges): same ABI entry, larger problems -------------------
@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("M,N,Kd,act", [(14400, 1024, 512, L.ACT_GELU), (3300, 1000, 192, L.ACT_NONE),
                                        (7201, 512, 1024, L.ACT_RELU), (25000, 151, 64, L.ACT_SILU)])
def test_gemm_tile_large_store(dt, M, N, Kd, act):
    A = rnd(M, Kd, seed=11).to(T(dt))
    W = (rnd(N, Kd, seed=12) / math.sqrt(Kd)).to(T(dt))
    bias = rnd(N, seed=13)
    ref = act_ref(A.double() @ W.double().T + bias.double(), act)
    for mode, odt in ((L.EPI_STORE_F32, torch.float32), (L.EPI_STORE_T, T(dt))):
        ldc = N + (8 if mode == L.EPI_STORE_T else 5)
        out = torch.full((M, ldc), 7.0, device=DEV, dtype=odt)
        K.gemm_tile(dt, A, W, M, N, Kd, bias=bias, act=act, mode=mode, out=out, ldc=ldc)
        torch.cuda.synchronize()
        assert relerr(out[:, :N], ref) < tol(dt), (mode, relerr(out[:, :N], ref))
        assert bool((out[:, N:] == 7.0).all()), "wrote outside the N columns"


@pytest.mark.parametrize("dt", DTS)
def test_gemm_tile_large_equals_row_pieces_bitwise(dt):
    """An output element does not depend on where its tile sits in the grid (XCD remap, ragged last panel)."""
    M, N, Kd = 14400, 1536, 512
    A = rnd(M, Kd, seed=14).to(T(dt))
    W = (rnd(N, Kd, seed=15) / math.sqrt(Kd)).to(T(dt))
    out = torch.empty(M, N, device=DEV, dtype=T(dt))
    K.gemm_tile(dt, A, W, M, N, Kd, act=L.ACT_GELU, mode=L.EPI_STORE_T, out=out, ldc=N)
    parts = []
    for lo in range(0, M, 1200):
        o = torch.empty(1200, N, device=DEV, dtype=T(dt))
        K.gemm_tile(dt, A[lo:lo + 1200], W, 1200, N, Kd, act=L.ACT_GELU, mode=L.EPI_STORE_T, out=o, ldc=N)
        parts.append(o)
    assert torch.equal(out, torch.cat(parts))


@pytest.mark.parametrize("dt", DTS)
def test_gemm_tile_large_split_amod_qkv_heads(dt):
    Lq, nseq, H, Lp = 450, 32, 8, 512
    M, Kd = nseq * Lq, 512
    A1, A2 = rnd(M, Kd, seed=16).to(T(dt)), rnd(M, Kd, seed=17).to(T(dt))
    W = (rnd(1536, Kd, seed=18) / math.sqrt(Kd)).to(T(dt))
    Q = torch.zeros(nseq + 1, H, Lp, 64, device=DEV, dtype=T(dt))
    Kk, Vv = torch.zeros_like(Q), torch.zeros_like(Q)
    K.gemm_tile(dt, A1, W, M, 1536, Kd, A2=A2, split_n=1024, mode=L.EPI_QKV_HEADS, out=Q, out_k=Kk,
                out_v=Vv, scale_q=0.125, Lseq=Lq, Lp=Lp, H=H, n_q=512, n_k=512, tok_off=1, seq_off=1)
    q = ((A1.double() @ W.double()[:512].T) * 0.125).view(nseq, Lq, H, 64).permute(0, 2, 1, 3)
    k = (A1.double() @ W.double()[512:1024].T).view(nseq, Lq, H, 64).permute(0, 2, 1, 3)
    v = (A2.double() @ W.double()[1024:].T).view(nseq, Lq, H, 64).permute(0, 2, 1, 3)
    assert relerr(Q[1:, :, 1:1 + Lq], q) < tol(dt)
    assert relerr(Kk[1:, :, 1:1 + Lq], k) < tol(dt)
    assert relerr(Vv[1:, :, 1:1 + Lq], v) < tol(dt)
    assert float(Q[0].abs().max()) == 0 and float(Vv[1:, :, 1 + Lq:].abs().max()) == 0 and float(Q[1:, :, :1].abs().max()) == 0
    # a_mod: 450 rows of A broadcast over 40 repeats
    out = torch.zeros(18000, 512, device=DEV)
    K.gemm_tile(dt, A1[:450], W[:512], 18000, 512, Kd, a_mod=450, mode=L.EPI_STORE_F32, out=out, ldc=512)
    ref = (A1[:450].double() @ W[:512].double().T).repeat(40, 1)
    assert relerr(out, ref) < tol(dt)


def rope_ref(n_pos, freqs):
    ang = torch.arange(n_pos, device=DEV, dtype=torch.float32)[:, None] * freqs[None, :]
    return torch.stack((ang.cos(), ang.sin()), -1).reshape(n_pos, 512)


def rot_ref(u, pos, freqs):
    ang = (pos.float()[:, None] * freqs[None, :]).repeat_interleave(2, -1).double()
    up = u.double().reshape(u.shape[0], -1, 2)
    rh = torch.stack((-up[..., 1], up[..., 0]), -1).reshape(u.shape)
    return u.double() * ang.cos() + rh * ang.sin()


def freqs512():
    return (1.0 / (10000 ** (torch.arange(0, 512, 2).float() / 512))).to(DEV)


def test_rope_table():
    fr = freqs512()
    rope = torch.empty(460, 512, device=DEV)
    K.rope_table(fr, rope, 460)
    assert float((rope - rope_ref(460, fr)).abs().max()) < 5e-6


def ln_ref(x, g, b, eps):
    x = x.double()
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * g.double() + b.double()


@pytest.mark.parametrize("dt", DTS)
def test_ln_rot(dt):
    rows, Lq = 333, 50
    x = rnd(rows, 512, seed=11, scale=3.0) + 0.7
    g, b = 1 + 0.1 * rnd(512, seed=12), 0.1 * rnd(512, seed=13)
    fr = freqs512()
    rope = torch.empty(64, 512, device=DEV)
    K.rope_table(fr, rope, 64)
    h = torch.zeros(rows, 512, device=DEV, dtype=T(dt))
    r = torch.zeros_like(h)
    y = torch.zeros(rows, 512, device=DEV)
    K.ln_rot(dt, x, rows, g, b, 1e-5, h=h, rot=r, y32=y, rope=rope, pos_mod=Lq, pos_base=3)
    u = ln_ref(x, g, b, 1e-5)
    assert relerr(y, u) < 1e-5
    assert relerr(h, u) < tol(dt, 1e-5, 1e-2)
    pos = torch.arange(rows, device=DEV) % Lq + 3
    assert relerr(r, rot_ref(u.float(), pos, fr)) < tol(dt, 1e-5, 1e-2)


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("Kd", [512, 1024])
def test_gemm_rowln_full_epilogue(dt, Kd):
    """fc -> LayerNorm(1e-6) -> FiLM -> residual -> store x; next LayerNorm(1e-5) -> h and rotary(h)."""
    Lq, nseq = 90, 3
    M = nseq * Lq  # 270: not a multiple of 64
    A = rnd(M, Kd, seed=20).to(T(dt))
    W = (rnd(512, Kd, seed=21) / math.sqrt(Kd)).to(T(dt))
    bias = rnd(512, seed=22)
    g1, b1 = 1 + 0.1 * rnd(512, seed=23), 0.1 * rnd(512, seed=24)
    g2, b2 = 1 + 0.1 * rnd(512, seed=25), 0.1 * rnd(512, seed=26)
    film = rnd(nseq, 3000, seed=27, scale=0.5)
    xres = rnd(M, 512, seed=28)
    fr = freqs512()
    rope = torch.empty(Lq, 512, device=DEV)
    K.rope_table(fr, rope, Lq)
    xout = torch.zeros(M, 512, device=DEV)
    h = torch.zeros(M, 512, device=DEV, dtype=T(dt))
    r = torch.zeros_like(h)
    K.gemm_rowln(dt, A, W, M, Kd, bias=bias, ln_g=g1, ln_b=b1, ln_eps=1e-6, film=film[:, 100:], film_ld=3000, xres=xres,
                 xout=xout, Lseq=Lq, nln_g=g2, nln_b=b2, nln_eps=1e-5, hout=h, rout=r, rope=rope,
                 flags=L.ROW_BIAS | L.ROW_LN_POST | L.ROW_FILM | L.ROW_STORE_X | L.ROW_NEXT_LN | L.ROW_STORE_H | L.ROW_STORE_ROT)
    v = ln_ref(A.double() @ W.double().T + bias.double(), g1, b1, 1e-6)
    seq = torch.arange(M, device=DEV) // Lq
    sc, sh = film[seq, 100:612].double(), film[seq, 612:1124].double()
    xr = xres.double() + (sc + 1) * v + sh
    assert relerr(xout, xr) < tol(dt, 2e-5, 5e-3), relerr(xout, xr)
    u = ln_ref(xr, g2, b2, 1e-5)
    assert relerr(h, u) < tol(dt, 2e-5, 1e-2)
    pos = torch.arange(M, device=DEV) % Lq
    assert relerr(r, rot_ref(u.float(), pos, fr)) < tol(dt, 2e-5, 1e-2)


@pytest.mark.parametrize("dt", DTS)
def test_gemm_rowln_variants(dt):
    M, Kd = 200, 512
    A = rnd(300, Kd, seed=30).to(T(dt))
    W = (rnd(512, Kd, seed=31) / math.sqrt(Kd)).to(T(dt))
    bias = rnd(512, seed=32)
    xres = rnd(100, 512, seed=33)
    acc = A.double() @ W.double().T + bias.double()
    # bias + residual with modulo rows (shared layer-0 input) + plain T copy
    xout = torch.zeros(M, 512, device=DEV)
    h = torch.zeros(M, 512, device=DEV, dtype=T(dt))
    K.gemm_rowln(dt, A, W, M, Kd, bias=bias, xres=xres, xres_mod=100, a_mod=150, xout=xout, hout=h, Lseq=50,
                 flags=L.ROW_BIAS | L.ROW_RES | L.ROW_STORE_X | L.ROW_STORE_H)
    idx = torch.arange(M, device=DEV)
    ref = xres.double()[idx % 100] + acc[idx % 150]
    assert relerr(xout, ref) < tol(dt, 2e-5, 5e-3)
    assert relerr(h, ref) < tol(dt, 2e-5, 1e-2)
    # de-interleaving output rows (fusion projection): m -> m*3 + 1
    xo = torch.zeros(3 * M, 512, device=DEV)
    K.gemm_rowln(dt, A, W, M, Kd, bias=bias, xout=xo, Lseq=50, out_mul=3, out_add=1, flags=L.ROW_BIAS | L.ROW_STORE_X)
    assert relerr(xo[1::3], acc[:M]) < tol(dt, 2e-5, 5e-3)
    assert float(xo[0::3].abs().max()) == 0 and float(xo[2::3].abs().max()) == 0
    # grouped launch (all dancers' slices of the fusion projection at once) == one launch per group, bit for bit
    G = 3
    Wg = (rnd(G * 512, Kd, seed=34) / math.sqrt(Kd)).to(T(dt))
    bg = rnd(G * 512, seed=35)
    g2, b2 = 1 + 0.1 * rnd(512, seed=36), 0.1 * rnd(512, seed=37)
    rope = rope_ref(60, freqs512())
    kw = dict(Lseq=60, flags=L.ROW_BIAS | L.ROW_STORE_X | L.ROW_NEXT_LN | L.ROW_STORE_H | L.ROW_STORE_ROT, nln_g=g2, nln_b=b2,
              nln_eps=1e-5, rope=rope, out_mul=G)
    outs = []
    for grouped in (True, False):
        xg = torch.zeros(G * M, 512, device=DEV)
        hg = torch.zeros(G * M, 512, device=DEV, dtype=T(dt))
        rg = torch.zeros(G * M, 512, device=DEV, dtype=T(dt))
        if grouped:
            K.gemm_rowln(dt, A, Wg, M, Kd, bias=bg, xout=xg, hout=hg, rout=rg, out_add=0, groups=G, **kw)
        else:
            for g in range(G):
                K.gemm_rowln(dt, A, Wg[g * 512:], M, Kd, bias=bg[g * 512:], xout=xg, hout=hg, rout=rg, out_add=g, **kw)
        outs.append((xg, hg, rg))
    for a_, b_ in zip(*outs):
        assert torch.equal(a_, b_)
    accg = A.double()[:M] @ Wg.double().T + bg.double()
    assert relerr(outs[0][0].view(M, G, 512), accg.view(M, G, 512)) < tol(dt, 2e-5, 5e-3)


def attn_ref(q, k, v):
    s = q.double() @ k.double().transpose(-1, -2)
    return torch.softmax(s, -1) @ v.double()


@pytest.mark.parametrize("dt", DTS)
@pytest.mark.parametrize("Lq,Lk,nseq,n_shared", [(450, 450, 3, 0), (450, 152, 4, 2), (120, 62, 2, 1), (150, 150, 2, 0),
                                                 (300, 1500, 1, 0), (1500, 1500, 2, 0), (1500, 302, 3, 1),
                                                 (700, 577, 2, 0), (513, 1025, 1, 0)])
def test_attention(dt, Lq, Lk, nseq, n_shared):
    H = 8
    Lpq, Lpk = K.round_up(Lq, 128), K.round_up(Lk, 128)
    n_kv = nseq if n_shared == 0 else nseq - n_shared + 1
    q = (rnd(nseq, H, Lq, 64, seed=40) * 0.5).to(T(dt))
    k = rnd(n_kv, H, Lk, 64, seed=41).to(T(dt))
    v = rnd(n_kv, H, Lk, 64, seed=42).to(T(dt))
    Q = torch.zeros(nseq, H, Lpq, 64, device=DEV, dtype=T(dt))
    Kk = torch.zeros(n_kv, H, Lpk, 64, device=DEV, dtype=T(dt))
    Vv = torch.zeros_like(Kk)
    Q[:, :, :Lq] = q
    Kk[:, :, :Lk] = k
    Vv[:, :, :Lk] = v
    O = torch.zeros(nseq * Lq, 512, device=DEV, dtype=T(dt))
    K.attention(dt, Q, Kk, Vv, O, nseq, H, Lq, Lk, Lpq, Lpk, 512, n_shared=n_shared)
    kv = torch.tensor([0 if s < n_shared else s - n_shared + (1 if n_shared > 0 else 0) for s in range(nseq)], device=DEV)
    ref = attn_ref(q, k[kv], v[kv]).permute(0, 2, 1, 3).reshape(nseq * Lq, 512)
    err = float((O.double() - ref).abs().max())
    assert err < tol(dt, 2e-5, 2e-2), err


@pytest.mark.parametrize("ng", [1, 2])
@pytest.mark.parametrize("Lq,Lk,nseq,n_shared", [(450, 450, 3, 0), (450, 152, 4, 2), (1500, 1500, 2, 0), (1500, 302, 3, 1),
                                                 (513, 1025, 1, 0)])
def test_attention_resident_kernel_both_row_groupings(ng, Lq, Lk, nseq, n_shared):
    """the K/V-resident bf16 kernel with 32 (ng=1) and 64 (ng=2) query rows per wave FORCED, including the multi-chunk
    key loop (L = 1500 > 512 keys per LDS-resident chunk): the launcher picks ng from the launch size, so the
    benchmarked B = 16 shape (ng = 1) and config 4 at B >= 3 (ng = 2 with chunks) both need explicit coverage."""
    dt, H = L.DT_BF16, 8
    Lpq, Lpk = K.round_up(Lq, 128), K.round_up(Lk, 128)
    if Lpq < 512:
        pytest.skip("resident kernel needs a Q image padded to >= 512 rows")
    n_kv = nseq if n_shared == 0 else nseq - n_shared + 1
    q = (rnd(nseq, H, Lq, 64, seed=46) * 0.5).to(T(dt))
    k = rnd(n_kv, H, Lk, 64, seed=47).to(T(dt))
    v = rnd(n_kv, H, Lk, 64, seed=48).to(T(dt))
    Q = torch.zeros(nseq, H, Lpq, 64, device=DEV, dtype=T(dt))
    Kk = torch.zeros(n_kv, H, Lpk, 64, device=DEV, dtype=T(dt))
    Vv = torch.zeros_like(Kk)
    Q[:, :, :Lq], Kk[:, :, :Lk], Vv[:, :, :Lk] = q, k, v
    O = torch.full((nseq * Lq, 512), 7.0, device=DEV, dtype=T(dt))
    K.attention(dt, Q, Kk, Vv, O, nseq, H, Lq, Lk, Lpq, Lpk, 512, n_shared=n_shared, ng=ng)
    kv = torch.tensor([0 if s < n_shared else s - n_shared + (1 if n_shared > 0 else 0) for s in range(nseq)], device=DEV)
    ref = attn_ref(q, k[kv], v[kv]).permute(0, 2, 1, 3).reshape(nseq * Lq, 512)
    err = float((O.double() - ref).abs().max())
    assert err < 2e-2, err
    # the two groupings only differ in which wave owns a row: same per-row arithmetic
    O2 = torch.zeros_like(O)
    K.attention(dt, Q, Kk, Vv, O2, nseq, H, Lq, Lk, Lpq, Lpk, 512, n_shared=n_shared, ng=3 - ng)
    assert float((O.float() - O2.float()).abs().max()) < 4e-3


def test_attention_large_logits_online_softmax():
    """force the running-max rescale branch: one key far above the rest, placed in a late tile"""
    dt, H, Lq, Lk = L.DT_F32, 8, 128, 200
    q = rnd(1, H, Lq, 64, seed=43)
    k = rnd(1, H, Lk, 64, seed=44)
    k[:, :, 170] = 4.0 * q[:, :, 5]
    v = rnd(1, H, Lk, 64, seed=45)
    Q = torch.zeros(1, H, 128, 64, device=DEV)
    Kk = torch.zeros(1, H, 256, 64, device=DEV)
    Vv = torch.zeros(1, H, 256, 64, device=DEV)
    Q[:, :, :Lq], Kk[:, :, :Lk], Vv[:, :, :Lk] = q, k, v
    O = torch.zeros(Lq, 512, device=DEV)
    K.attention(dt, Q, Kk, Vv, O, 1, H, Lq, Lk, 128, 256, 512)
    ref = attn_ref(q, k, v).permute(0, 2, 1, 3).reshape(Lq, 512)
    assert float((O.double() - ref).abs().max()) < 5e-5


@pytest.mark.parametrize("dt", DTS)
def test_small_ops(dt):
    # convert_pad with batch/row strides (music frame pairing, model/model.py:572-576)
    B, S, Cd = 3, 10, 438
    cond = rnd(B, 2 * S + 1, Cd, seed=50)
    dst = torch.full((B * S, 896), 9.0, device=DEV, dtype=T(dt))
    K.convert_pad(dt, cond, dst, B * S, 2 * Cd, 896, rows_per_batch=S, batch_stride=(2 * S + 1) * Cd, row_stride=2 * Cd)
    ref = cond[:, :-1].reshape(B * S, 2 * Cd)
    assert relerr(dst[:, :876], ref) < tol(dt, 1e-7, 5e-3) and float(dst[:, 876:].abs().max()) == 0
    # sinusoidal
    times = torch.tensor([0, 1, 37, 999], dtype=torch.int32, device=DEV)
    f = torch.exp(torch.arange(256) * -(math.log(10000) / 255)).to(DEV)
    emb = torch.zeros(4, 512, device=DEV, dtype=T(dt))
    K.sinusoidal(dt, times, 4, f, emb)
    e = times.float()[:, None] * f[None]
    assert relerr(emb, torch.cat((e.sin(), e.cos()), -1)) < tol(dt, 2e-6, 5e-3)
    # mean pool
    x = rnd(B, S, 512, seed=51)
    out = torch.zeros(B, 512, device=DEV)
    K.mean_pool(x, out, B, S, 512)
    assert relerr(out, x.mean(1)) < 1e-6
    # add + mish
    a, bb = rnd(5, 512, seed=52), rnd(4, 512, seed=53)
    ia = torch.tensor([4, 0, 0, 2], dtype=torch.int32, device=DEV)
    o = torch.zeros(4, 512, device=DEV, dtype=T(dt))
    o32 = torch.zeros(4, 512, device=DEV)
    K.add_act(dt, a, ia, bb, 4, L.ACT_MISH, out=o, out32=o32)
    ref = torch.nn.functional.mish(a[ia.long()] + bb)
    assert relerr(o32, ref) < 2e-6 and relerr(o, ref) < tol(dt, 2e-6, 5e-3)
    # cfg combine
    ou, oc = rnd(7, 152, seed=54), rnd(7, 152, seed=55)
    y = torch.zeros(7, 151, device=DEV)
    K.cfg_combine(ou, oc, 152, 2.0, y, 7, 151)
    assert relerr(y, ou[:, :151] + (oc[:, :151] - ou[:, :151]) * 2.0) < 1e-6
    # window coupling
    xw = rnd(3, 6, 10, seed=56)
    exp = xw.clone()
    exp[1:, :3] = xw[:-1, 3:]
    K.window_couple(xw, 3, 6, 10)
    assert torch.equal(xw, exp)


@pytest.mark.parametrize("dt", DTS)
def test_scatter_time_kv(dt):
    NL, n_t, n_kv, H, Lp, S = 2, 5, 3, 8, 128, 60
    tab = rnd(NL, n_t, 2, 1024, seed=60).to(T(dt))
    tidx = torch.tensor([4, 0, 2], dtype=torch.int32, device=DEV)
    Kc = torch.zeros(NL, n_kv, H, Lp, 64, device=DEV, dtype=T(dt))
    Vc = torch.zeros_like(Kc)
    K.scatter_time_kv(dt, tab, n_t, tidx, Kc, Vc, NL, n_kv, H, Lp, S)
    for l in range(NL):
        for s in range(n_kv):
            for r in range(2):
                row = tab[l, tidx[s].item(), r]
                assert torch.equal(Kc[l, s, :, S + r].reshape(-1), row[:512])
                assert torch.equal(Vc[l, s, :, S + r].reshape(-1), row[512:])
    assert float(Kc[:, :, :, :S].abs().max()) == 0


def test_sampler_update_ddpm_ddim_and_philox():
    rows, nf, Lq = 240, 151, 120
    ou, oc = rnd(rows, 152, seed=70), rnd(rows, 152, seed=71)
    x = rnd(rows, nf, seed=72)
    eps = rnd(rows, nf, seed=73)
    traj = rnd(rows, 3, seed=74)
    counter = torch.tensor([1, 0, 0, 0], dtype=torch.int32, device=DEV)
    tseq = torch.tensor([9, 8], dtype=torch.int32, device=DEV)
    params = torch.tensor([[0] * 8, [2.0, 0.3, 0.6, 0.2, 0.5, 0.1, 0.0, 0]], device=DEV)
    g = ou[:, :nf] + (oc[:, :nf] - ou[:, :nf]) * 2.0
    x0 = g.clamp(-1, 1)
    # DDPM
    x1 = x.clone()
    x0o = torch.zeros_like(x)
    K.sampler_update(L.SAMPLER_DDPM, ou, oc, 152, x1, eps, None, x0o, rows, nf, Lq, counter, params, tseq)
    assert relerr(x1, (0.3 * x0 + 0.6 * x) + 0.2 * eps) < 1e-6 and relerr(x0o, x0) < 1e-7
    # DDPM, cond-only branch
    x1 = x.clone()
    K.sampler_update(L.SAMPLER_DDPM, None, oc, 152, x1, eps, None, None, rows, nf, Lq, counter, params, tseq)
    assert relerr(x1, (0.3 * oc[:, :nf].clamp(-1, 1) + 0.6 * x) + 0.2 * eps) < 1e-6
    # DDIM with trajectory overwrite
    x1 = x.clone()
    K.sampler_update(L.SAMPLER_DDIM, ou, oc, 152, x1, eps, traj, None, rows, nf, Lq, counter, params, tseq)
    pn = (0.3 * x - x0) / 0.6
    ref = (x0 * 0.2 + 0.5 * pn) + 0.1 * eps
    ref[:, 4:6] = traj[:, 0:2]
    assert relerr(x1, ref) < 1e-6
    # DDIM last step -> x0
    params[1, 6] = 1.0
    x1 = x.clone()
    K.sampler_update(L.SAMPLER_DDIM, ou, oc, 152, x1, eps, None, None, rows, nf, Lq, counter, params, tseq)
    assert relerr(x1, x0) < 1e-7
    # Philox noise: N(0,1) statistics, reproducible, keyed by global clip index (partition invariance)
    params[1] = torch.tensor([1.0, 0.0, 0.0, 1.0, 0, 0, 0, 0], device=DEV)
    z = torch.zeros(rows, nf, device=DEV)
    K.sampler_update(L.SAMPLER_DDPM, None, oc, 152, z, None, None, None, rows, nf, Lq, counter, params, tseq, seed=123, clip0=4)
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02
    z2 = torch.zeros(Lq, nf, device=DEV)
    K.sampler_update(L.SAMPLER_DDPM, None, oc, 152, z2, None, None, None, Lq, nf, Lq, counter, params, tseq, seed=123, clip0=5)
    assert torch.equal(z2, z[Lq:])          # clip 5 alone == second clip of the (4,5) batch
    z3 = torch.zeros(rows, nf, device=DEV)
    K.sampler_update(L.SAMPLER_DDPM, None, oc, 152, z3, None, None, None, rows, nf, Lq, counter, params, tseq, seed=124, clip0=4)
    assert not torch.equal(z3, z)


def test_step_counter():
    counter = torch.zeros(4, dtype=torch.int32, device=DEV)
    tseq = torch.tensor([7, 5, 3], dtype=torch.int32, device=DEV)
    tidx = torch.zeros(6, dtype=torch.int32, device=DEV)
    for want in (7, 5, 3):
        K.step_begin(counter, tseq, tidx, 6)
        assert tidx.tolist() == [want] * 6
        K.step_end(counter)
    assert counter[0].item() == 3


@pytest.mark.parametrize("dt,frags", [(L.DT_F32, False), (L.DT_BF16, False), (L.DT_BF16, True)])
def test_step_prologue_equals_the_separate_launches(dt, frags):
    """tcdiff_step_prologue == step_begin + add_act(mish) + scatter_time_kv (+ pack_kv_frags) + convert_pad + step_end,
    bit for bit, over three consecutive steps (the counter protocol: counter[0] current, counter[3] next, advanced by sampler_update | SAMPLER_ADVANCE)."""
    NL, n_kv, H, S, n_t, n_seq, rows, nf = 3, 6, 8, 40, 5, 6, 130, 151
    Lp, nkt = 128, (S + 2 + 31) // 32
    tdt = T(dt)
    tab = rnd(NL, n_t, 2, 1024, seed=1).to(tdt)
    t_base, hidden = rnd(n_t, 512, seed=2), rnd(n_seq, 512, seed=3)
    tseq = torch.tensor([4, 2, 0], dtype=torch.int32, device=DEV)
    x = rnd(rows, nf, seed=4)

    def bufs():
        g = torch.Generator(device="cpu").manual_seed(9)
        mk = lambda *sh: torch.randn(*sh, generator=g).to(DEV).to(tdt)      # noqa: E731
        return dict(Kc=mk(NL, n_kv, H, Lp, 64), Vc=mk(NL, n_kv, H, Lp, 64), Kf=mk(NL, n_kv, H, nkt * 2048),
                    Vf=mk(NL, n_kv, H, nkt * 2048), film_in=torch.zeros(n_seq, 512, device=DEV, dtype=tdt),
                    xin=torch.full((rows, 192), 7.0, device=DEV, dtype=tdt),
                    tidx=torch.zeros(n_seq, dtype=torch.int32, device=DEV),
                    counter=torch.zeros(8, dtype=torch.int32, device=DEV))
    a, b = bufs(), bufs()
    for step in range(3):
        K.step_begin(a["counter"], tseq, a["tidx"], n_seq)
        K.add_act(dt, t_base, a["tidx"], hidden, n_seq, L.ACT_MISH, out=a["film_in"])
        K.scatter_time_kv(dt, tab, n_t, a["tidx"], a["Kc"], a["Vc"], NL, n_kv, H, Lp, S)
        if frags:
            K.pack_kv_frags(a["Kc"], a["Vc"], a["Kf"], a["Vf"], NL * n_kv, H, Lp, nkt, S, S + 2)
        K.convert_pad(dt, x, a["xin"], rows, nf, 192)
        K.step_end(a["counter"])
        K.step_prologue(dt, b["counter"], tseq, b["tidx"], t_base, hidden, b["film_in"], n_seq, tab, n_t,
                        None if frags else b["Kc"], None if frags else b["Vc"], b["Kf"] if frags else None,
                        b["Vf"] if frags else None, NL, n_kv, H, Lp, nkt if frags else 0, S, x, b["xin"], rows, nf, 192)
        dummy = torch.zeros(4, 152, device=DEV)
        K.sampler_update(L.SAMPLER_DDPM | L.SAMPLER_ADVANCE, None, dummy, 152, torch.zeros(4, 151, device=DEV), None, None,
                         None, 4, 151, 4, b["counter"], torch.zeros(3, 8, device=DEV), tseq)
        torch.cuda.synchronize()
        assert b["counter"][0].item() == step and b["counter"][3].item() == step + 1
        assert a["counter"][0].item() == step + 1
        names = ["film_in", "xin", "tidx"] + (["Kf", "Vf"] if frags else ["Kc", "Vc"])
        for n in names:
            assert torch.equal(a[n], b[n]), (n, step)
        x = x * 0.5 + 0.1


def test_argument_errors():
    a = torch.zeros(64, 64, device=DEV)
    with pytest.raises(L.TcdiffError):
        K.gemm_tile(L.DT_F32, a, a, 64, 64, 60, mode=L.EPI_STORE_F32, out=a, ldc=64)      # K not a multiple of 32
    with pytest.raises(L.TcdiffError):
        K.attention(L.DT_F32, a, a, a, a, 1, 8, 10, 10, 100, 128, 512)                      # Lp_q not a multiple of 128
    with pytest.raises(L.TcdiffError):
        K.gemm_rowln(L.DT_F32, a, a, 64, 64, flags=L.ROW_BIAS, Lseq=1)                      # bias flag without bias


def test_ema_update_matches_reference_rounding():
    """model/diffusion.py:61-76: ma = ma * beta + (1 - beta) * cur, bit for bit, ragged sizes, one launch."""
    from tcdiff_amd.diffusion import EMA
    sizes = [(512, 512), (151,), (1024, 1536), (3,), (70001,)]
    cur = [torch.nn.Parameter(rnd(*s, seed=90 + i)) for i, s in enumerate(sizes)]
    ma = [torch.nn.Parameter(rnd(*s, seed=190 + i)) for i, s in enumerate(sizes)]

    class Bag(torch.nn.Module):
        def __init__(self, ps):
            super().__init__()
            self.ps = torch.nn.ParameterList(ps)

    ref = [m.data * 0.9999 + (1 - 0.9999) * c.data for m, c in zip(ma, cur)]
    ema = EMA(0.9999)
    ema.update_model_average(Bag(ma), Bag(cur))
    torch.cuda.synchronize()
    for m, r in zip(ma, ref):
        assert torch.equal(m.data, r)
    ref2 = [r * 0.9999 + (1 - 0.9999) * c.data for r, c in zip(ref, cur)]
    ema.update_model_average(Bag(ma), Bag(cur))      # cached chunk table
    for m, r in zip(ma, ref2):
        assert torch.equal(m.data, r)

