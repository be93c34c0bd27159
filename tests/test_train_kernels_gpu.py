"""Kernel-level checks of the training-step launchers (through the C ABI) against plain torch evaluations of the same
operator and torch autograd of it: operand repacking, split-K weight gradients, activation + dropout, the row-local block
glue (forward and reverse), train-mode attention (forward and reverse) and the loss terms' reverse pass.  Dropout masks
come from the oracle's numpy evaluation of the product's counter hash, so a mismatch in the hash shows up here first."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import tcdiff_oracle as O  # noqa: E402  (checker only)
from tcdiff_amd import _lib as L  # noqa: E402
from tcdiff_amd import kernels as K  # noqa: E402

DEV = "cuda"
SEED = (1234, 5678)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def seed_dev():
    return torch.tensor(SEED, dtype=torch.int32, device=DEV)


def mode(compute):
    dt = K.dtype_id(compute)
    return dt, K.TORCH_DT[dt], (2e-6 if compute == "f32" else 1.5e-2)


@pytest.mark.parametrize("compute", ["f32", "bf16"])
@pytest.mark.parametrize("src_f32", [True, False])
def test_cast_transpose_pads_and_column_sums(compute, src_f32):
    dt, T, _ = mode(compute)
    rows, cols, ld = 150, 151, 152
    g = torch.Generator().manual_seed(1)
    src = torch.randn(rows, ld, generator=g)
    srcd = src.to(DEV) if src_f32 else src.to(DEV, T)
    ref = srcd.float().cpu()[:, :cols]
    cp, rp = 192, 192
    dst = torch.full((rows, cp), 7.0, device=DEV, dtype=T)
    dstT = torch.full((cols, rp), 7.0, device=DEV, dtype=T)
    cs = torch.zeros(cols, device=DEV)
    K.cast_transpose(dt, srcd, rows, cols, ld, dst=dst, ld_dst=cp, cols_pad=cp, dstT=dstT, ld_dstT=rp, rows_pad=rp, colsum=cs)
    want = ref.to(T).float()
    assert torch.equal(dst.float().cpu()[:, :cols], want) and float(dst.float()[:, cols:].abs().max()) == 0.0
    assert torch.equal(dstT.float().cpu()[:, :rows], want.t()) and float(dstT.float()[:, rows:].abs().max()) == 0.0
    assert rel(cs, ref.sum(0)) < 1e-6


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_gemm_splitk_accumulates_the_weight_gradient(compute):
    dt, T, tol = mode(compute)
    kt = K.k_tile(dt)
    n_out, n_in, rows = 200, 151, 1000                       # dW[n_out, n_in] += dY^T X over `rows` token rows
    rp = K.round_up(rows, kt)
    g = torch.Generator().manual_seed(2)
    dYt = torch.zeros(n_out, rp, dtype=T)
    Xt = torch.zeros(n_in, rp, dtype=T)
    dYt[:, :rows] = torch.randn(n_out, rows, generator=g).to(T)
    Xt[:, :rows] = torch.randn(n_in, rows, generator=g).to(T)
    init = torch.randn(n_out, n_in, generator=g)
    out = init.clone().to(DEV)
    for splits in (1, 5):
        out.copy_(init)
        K.gemm_splitk(dt, dYt.to(DEV), Xt.to(DEV), n_out, n_in, rp, rp, rp, out, n_in, splits)
        want = init.double() + dYt.double() @ Xt.double().t()
        assert rel(out, want) < (1e-6 if compute == "f32" else 1e-5), splits


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_gemm_tn_weight_gradient_from_token_major_operands(compute):
    """dW[n_out, n_in] += dY^T X with dY and X as the forward left them (token-major, leading dimensions wider than the
    slices used: the fused Q|K|V gradient is read in place)."""
    dt, T, tol = mode(compute)
    n_out, n_in, rows, ld_dy, ldx = 256, 384, 1344, 640, 512          # 1344 = 21 k-tiles of 64 (42 of 32)
    g = torch.Generator().manual_seed(12)
    dY = torch.randn(rows, ld_dy, generator=g).to(T)
    X = torch.randn(rows, ldx, generator=g).to(T)
    init = torch.randn(n_out, n_in, generator=g)
    c0 = 128                                                          # the slice dY[:, 128:384]
    want = init.double() + dY[:, c0:c0 + n_out].double().t() @ X[:, :n_in].double()
    out = init.clone().to(DEV)
    for splits in (1, 2, 7, 64):
        out.copy_(init)
        K.gemm_tn(dt, dY.to(DEV).view(-1)[c0:], X.to(DEV), n_out, n_in, rows, ld_dy, ldx, out, n_in, splits)
        assert rel(out, want) < (1e-6 if compute == "f32" else 1e-5), splits
    assert K.gemm_tn_ok(dt, n_out, n_in, rows) and not K.gemm_tn_ok(dt, 200, n_in, rows) and not K.gemm_tn_ok(dt, n_out, n_in, 1000)
    with pytest.raises(L.TcdiffError):                                 # unsupported shapes are refused, not rounded
        K.gemm_tn(dt, dY.to(DEV), X.to(DEV), 200, n_in, rows, ld_dy, ldx, out, n_in, 1)


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_gemm_tn_grouped_adds_every_weight_gradient_of_the_group(compute):
    """several dW_i += dY_i^T X_i in one launch whose (tile, k-tile) units are cut into equal shares: shares cross tile AND
    problem boundaries (different shapes, different token counts), every output still gets exactly its own sum"""
    dt, T, tol = mode(compute)
    g = torch.Generator().manual_seed(15)
    shapes = [(256, 128, 1344), (128, 384, 1344), (512, 512, 704), (128, 128, 64), (384, 256, 2112)]     # (n_out, n_in, tokens)
    probs, checks, dYs, Xs = [], [], [], []
    for n_out, n_in, rows in shapes:
        ld_dy, ldx = n_out + 128, n_in
        dY = torch.randn(rows, ld_dy, generator=g).to(T)
        X = torch.randn(rows, ldx, generator=g).to(T)
        init = torch.randn(n_out, n_in, generator=g)
        out = init.clone().to(DEV)
        want = init.double() + dY[:, 64:64 + n_out].double().t() @ X.double()
        dYd, Xd = dY.to(DEV), X.to(DEV)
        dYs.append(dY)
        Xs.append(X)
        probs.append((dYd.view(-1)[64:], Xd, n_out, n_in, rows, ld_dy, ldx, out, n_in))
        checks.append((out, want))
    K.gemm_tn_grouped(dt, probs)
    for i, (out, want) in enumerate(checks):
        assert rel(out, want) < (1e-6 if compute == "f32" else 1e-5), i
    K.gemm_tn_grouped(dt, probs[:1])                       # a group of one accumulates on top: init + 2 x product
    prod = dYs[0][:, 64:64 + shapes[0][0]].double().t() @ Xs[0].double()
    assert rel(checks[0][0], checks[0][1] + prod) < (1e-6 if compute == "f32" else 1e-5)
    with pytest.raises(L.TcdiffError):
        K.gemm_tn_grouped(dt, [(probs[0][0], probs[0][1], 200, 128, 1344, probs[0][5], probs[0][6], probs[0][7], 128)])


@pytest.mark.parametrize("compute", ["f32", "bf16"])
@pytest.mark.parametrize("act,p", [(L.ACT_GELU, 0.1), (L.ACT_RELU, 0.0), (L.ACT_SILU, 0.1)])
def test_gemm_tile_fused_activation_epilogues_equal_the_separate_launches(compute, act, p):
    """nn.Linear + activation + nn.Dropout in one launch (out = pre-activation, out2 = its activated, dropped image) and the
    input-gradient GEMM with the activation's backward in its epilogue: bit-identical to gemm_tile followed by act_drop(_bwd)."""
    dt, T, tol = mode(compute)
    M, N, Kd = 300, 1024, 512
    g = torch.Generator().manual_seed(14)
    A = torch.randn(M, Kd, generator=g).to(T).to(DEV)
    W = (torch.randn(N, Kd, generator=g) * 0.05).to(T).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    thr, sc = K.drop_params(p)
    site = 22
    a1, y1 = torch.empty(M, N, device=DEV, dtype=T), torch.empty(M, N, device=DEV, dtype=T)
    K.gemm_tile(dt, A, W, M, N, Kd, bias=bias, out=a1, ldc=N)
    K.act_drop(dt, a1, N, y1, N, M, N, act, seed_dev(), site, thr, sc)
    a2, y2 = torch.empty(M, N, device=DEV, dtype=T), torch.empty(M, N, device=DEV, dtype=T)
    K.gemm_tile(dt, A, W, M, N, Kd, bias=bias, out=a2, ldc=N, out2=y2, ldc2=N, act2=act, seed=seed_dev(), site=site, thr=thr,
                drop_scale=sc)
    assert torch.equal(a1, a2) and torch.equal(y1, y2)
    if p > 0:
        frac = float((y2 == 0).float().mean())
        assert 0.05 < frac < 0.6                                  # the mask is live (GELU / SiLU outputs are almost never exactly 0)
    # backward: dX = dY W2^T through the activation; dY [M, 512], W2T [1024 (k_in), 512]
    dY = torch.randn(M, 512, generator=g).to(T).to(DEV)
    W2T = (torch.randn(N, 512, generator=g) * 0.05).to(T).to(DEV)
    d1, da1 = torch.empty(M, N, device=DEV, dtype=T), torch.empty(M, N, device=DEV, dtype=T)
    K.gemm_tile(dt, dY, W2T, M, N, 512, out=d1, ldc=N)
    K.act_drop_bwd(dt, a1, N, d1, N, da1, M, N, act, seed_dev(), site, thr, sc)
    da2 = torch.empty(M, N, device=DEV, dtype=T)
    K.gemm_tile(dt, dY, W2T, M, N, 512, out=da2, ldc=N, act_src=a1, ld_src=N, act2=act, seed=seed_dev(), site=site, thr=thr,
                drop_scale=sc)
    assert torch.equal(da1, da2)
    with pytest.raises(L.TcdiffError):                            # ragged widths are refused (the engine launches act_drop instead)
        K.gemm_tile(dt, A, W, M, 438, Kd, out=a2, ldc=N, out2=y2, ldc2=N, act2=act)


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_cast_transpose_multi_equals_the_single_launches(compute):
    dt, T, tol = mode(compute)
    g = torch.Generator().manual_seed(13)
    shapes = [(512, 512), (200, 151), (64, 1024), (1, 70), (1536, 512)]
    srcs = [torch.randn(r, c, generator=g).to(DEV) for r, c in shapes]
    kt = K.k_tile(dt)
    ents, singles = [], []
    for w, (r, c) in zip(srcs, shapes):
        cp, rp = K.round_up(c, kt), K.round_up(r, kt)
        d1, t1 = torch.full((r, cp), 7.0, device=DEV, dtype=T), torch.full((c, rp), 7.0, device=DEV, dtype=T)
        d2, t2 = torch.full((r, cp), 9.0, device=DEV, dtype=T), torch.full((c, rp), 9.0, device=DEV, dtype=T)
        ents.append(dict(src=w, rows=r, cols=c, ld_src=c, dst=d1, ld_dst=cp, cols_pad=cp, dstT=t1, ld_dstT=rp, rows_pad=rp))
        K.cast_transpose(dt, w, r, c, c, dst=d2, ld_dst=cp, cols_pad=cp, dstT=t2, ld_dstT=rp, rows_pad=rp)
        singles.append((d1, t1, d2, t2))
    tab = K.ct_table(dt, ents, DEV)
    K.cast_transpose_multi(dt, tab)
    for (d1, t1, d2, t2), w, (r, c) in zip(singles, srcs, shapes):
        assert torch.equal(d1, d2) and torch.equal(t1, t2)
        assert torch.equal(d1[:, :c].float(), w.to(T).float()) and torch.equal(t1[:, :r].float(), w.t().to(T).float())
        assert float(d1[:, c:].abs().max() if d1.shape[1] > c else 0) == 0.0


@pytest.mark.parametrize("compute", ["f32", "bf16"])
@pytest.mark.parametrize("act", [L.ACT_RELU, L.ACT_GELU, L.ACT_MISH, L.ACT_SILU])
def test_act_drop_forward_and_backward(compute, act):
    dt, T, tol = mode(compute)
    rows, cols, ld = 37, 438, 448
    g = torch.Generator().manual_seed(3)
    a = (torch.randn(rows, ld, generator=g) * 2).to(T)
    dy = torch.randn(rows, ld, generator=g).to(T)
    thr, sc = K.drop_params(0.1)
    site = 22
    y = torch.full((rows, ld), 3.0, device=DEV, dtype=T)
    K.act_drop(dt, a.to(DEV), ld, y, ld, rows, cols, act, seed_dev(), site, thr, sc)
    keep = O.dropout_keep(SEED, site, (rows, cols), 0.1)
    fn = {L.ACT_RELU: F.relu, L.ACT_GELU: F.gelu, L.ACT_MISH: F.mish, L.ACT_SILU: F.silu}[act]
    ar = a[:, :cols].double().requires_grad_(True)
    yr = fn(ar) * keep / 0.9
    yr.backward(dy[:, :cols].double())
    assert rel(y[:, :cols], yr.detach()) < max(tol, 1e-6) and float(y[:, cols:].float().abs().max()) == 0.0
    # the same zero pattern: the hash is the oracle's
    assert torch.equal((y[:, :cols].float().cpu() != 0) | (yr.detach() == 0), torch.ones(rows, cols, dtype=torch.bool))
    da = torch.full((rows, ld), 3.0, device=DEV, dtype=T)
    K.act_drop_bwd(dt, a.to(DEV), ld, dy.to(DEV), ld, da, rows, cols, act, seed_dev(), site, thr, sc)
    assert rel(da[:, :cols], ar.grad) < max(tol, 1e-6) and float(da[:, cols:].float().abs().max()) == 0.0


def test_pos_drop_is_positional_encoding_in_train_mode():
    """tcdiff_pos_drop: x = dropout(x + pe[row % L]) in place on fp32 rows (PositionalEncoding.forward in train mode, model/utils.py:27-32
    at model/model.py:564,580) with the oracle's counter-hash mask of the site; without a table: dropout only; with threshold 0: the
    addition only.  Its backward is tcdiff_act_drop_bwd of TC_ACT_NONE at the same site (same flat index m * cols + c)."""
    rows, Lm, cols, site = 3 * 37, 37, 512, 9
    g = torch.Generator().manual_seed(5)
    x0, pe = torch.randn(rows, cols, generator=g), torch.randn(50, cols, generator=g)
    thr, sc = K.drop_params(0.1)
    keep = O.dropout_keep(SEED, site, (rows, cols), 0.1)
    want = (x0 + pe[:Lm].repeat(3, 1)) * keep * np.float32(1.0 / 0.9)
    x = x0.to(DEV).clone()
    K.pos_drop(x, rows, cols, pe.to(DEV), Lm, seed_dev(), site, thr, sc)
    assert torch.equal((x.cpu() != 0) | (want == 0), torch.ones(rows, cols, dtype=torch.bool)) and rel(x, want) < 1e-6
    x = x0.to(DEV).clone()
    K.pos_drop(x, rows, cols, None, 1, seed_dev(), site, thr, sc)
    assert rel(x, x0 * keep * np.float32(1.0 / 0.9)) < 1e-6
    x = x0.to(DEV).clone()
    K.pos_drop(x, rows, cols, pe.to(DEV), Lm)
    assert torch.equal(x.cpu(), x0 + pe[:Lm].repeat(3, 1))
    dy = torch.randn(rows, cols, generator=g).to(DEV)
    da = torch.empty_like(dy)
    K.act_drop_bwd(L.DT_F32, dy, cols, dy, cols, da, rows, cols, L.ACT_NONE, seed_dev(), site, thr, sc)
    assert rel(da, dy.cpu() * keep * np.float32(1.0 / 0.9)) < 1e-6
    with pytest.raises(L.TcdiffError):
        K.pos_drop(x, rows, 510, pe.to(DEV), Lm)            # whole 16-byte quads only


def _rope(n_pos):
    freqs = (1.0 / (10000 ** (torch.arange(0, 512, 2).float() / 512))).to(DEV)
    rope = torch.empty(n_pos, 512, device=DEV)
    K.rope_table(freqs, rope, n_pos)
    return rope


def _rot_ref(u, cs):
    up, cp = u.reshape(*u.shape[:-1], 256, 2), cs.reshape(*cs.shape[:-1], 256, 2)
    c, s = cp[..., 0], cp[..., 1]
    return torch.stack((up[..., 0] * c - up[..., 1] * s, up[..., 1] * c + up[..., 0] * s), -1).reshape(u.shape)


ROW_CASES = {
    "decoder self/cross block + rotary": L.ROWF_DROP_PRE | L.ROWF_LN_POST | L.ROWF_DROP_POST | L.ROWF_FILM | L.ROWF_RES | L.ROWF_STORE_X | L.ROWF_NEXT_LN | L.ROWF_STORE_H | L.ROWF_STORE_ROT,
    "decoder feed-forward block": L.ROWF_DROP_PRE | L.ROWF_FILM | L.ROWF_RES | L.ROWF_STORE_X | L.ROWF_NEXT_LN | L.ROWF_STORE_H,
    "norm + rotary only": L.ROWF_NEXT_LN | L.ROWF_STORE_H | L.ROWF_STORE_ROT,
    "encoder residual + dropout": L.ROWF_BIAS | L.ROWF_DROP_PRE | L.ROWF_RES | L.ROWF_STORE_X,
}


@pytest.mark.parametrize("compute", ["f32", "bf16"])
@pytest.mark.parametrize("case", list(ROW_CASES))
def test_row_block_forward_and_backward_vs_torch_autograd(compute, case):
    dt, T, tol = mode(compute)
    f = ROW_CASES[case]
    nseq, Ls = 3, 50
    M = nseq * Ls
    g = torch.Generator().manual_seed(4)
    rn = lambda *s: torch.randn(*s, generator=g)
    z, xres, bias = rn(M, 512), rn(M, 512), rn(512) * 0.3
    lg, lb, ng, nb = 1 + 0.1 * rn(512), 0.1 * rn(512), 1 + 0.1 * rn(512), 0.1 * rn(512)
    film = 0.3 * rn(nseq, 2048)                                      # block at column 1024 of a wider FiLM row
    rope = _rope(64)
    d_xn, d_h, d_r = rn(M, 512), rn(M, 512).to(T), rn(M, 512).to(T)
    thr, sc = K.drop_params(0.1)
    sp, sq = 17, 18
    dev = lambda t: t.to(DEV).contiguous()
    zd, xd = dev(z), dev(xres)
    xout, hout, rout = (torch.zeros(M, 512, device=DEV), torch.zeros(M, 512, device=DEV, dtype=T),
                        torch.zeros(M, 512, device=DEV, dtype=T))
    common = dict(flags=f, M=M, L=Ls, z=zd, bias=dev(bias), ln_g=dev(lg), ln_b=dev(lb), ln_eps=1e-6, film=dev(film)[:, 1024:],
                  film_ld=2048, xres=xd, nln_g=dev(ng), nln_b=dev(nb), nln_eps=1e-5, rope=rope, pos_mod=Ls, pos_base=3,
                  seed=seed_dev(), drop_thr=thr, drop_scale=sc, site_pre=sp, site_post=sq)
    keepers = dict(common)           # keep the device tensors alive
    K.row_fwd(dt, K.row_args(xout=xout, hout=hout, rout=rout, **common))
    # ---- torch reference (float64, autograd) ---------------------------------------------------------------------------
    D = torch.float64
    zr, xr, br = z.to(D).requires_grad_(True), xres.to(D).requires_grad_(True), bias.to(D).requires_grad_(True)
    lgr, lbr, ngr, nbr = (t.to(D).requires_grad_(True) for t in (lg, lb, ng, nb))
    fr = film.to(D).requires_grad_(True)
    v = zr
    if f & L.ROWF_BIAS:
        v = v + br
    if f & L.ROWF_DROP_PRE:
        v = v * O.dropout_keep(SEED, sp, (M, 512), 0.1) / 0.9
    if f & L.ROWF_LN_POST:
        v = F.layer_norm(v, (512,), lgr, lbr, 1e-6)
    if f & L.ROWF_DROP_POST:
        v = v * O.dropout_keep(SEED, sq, (M, 512), 0.1) / 0.9
    if f & L.ROWF_FILM:
        sc_, sh_ = fr[:, 1024:1536], fr[:, 1536:2048]
        v = ((sc_ + 1)[:, None, :] * v.view(nseq, Ls, 512) + sh_[:, None, :]).reshape(M, 512)
    if f & L.ROWF_RES:
        v = xr + v
    xn = v
    u = F.layer_norm(xn, (512,), ngr, nbr, 1e-5) if f & L.ROWF_NEXT_LN else xn
    pos = 3 + torch.arange(M) % Ls
    rot = _rot_ref(u, rope.cpu().to(D)[pos])
    loss = (xn * d_xn.to(D)).sum()
    if f & L.ROWF_STORE_H:
        loss = loss + (u * d_h.to(D)).sum()
    if f & L.ROWF_STORE_ROT:
        loss = loss + (rot * d_r.to(D)).sum()
    loss.backward()
    if f & L.ROWF_STORE_X:
        assert rel(xout, xn.detach()) < 2e-6
    if f & L.ROWF_STORE_H:
        assert rel(hout, u.detach()) < max(tol / 3, 2e-6)
    if f & L.ROWF_STORE_ROT:
        assert rel(rout, rot.detach()) < max(tol / 3, 2e-6)
    # ---- backward -----------------------------------------------------------------------------------------------------------
    chunks = 3
    part = torch.zeros(chunks * nseq, 5, 512, device=DEV)
    d_z = torch.zeros(M, 512, device=DEV, dtype=T)
    d_xres = torch.zeros(M, 512, device=DEV)
    d_film = torch.zeros(nseq, 2048, device=DEV)
    K.row_bwd(dt, K.row_args(d_xn=dev(d_xn), d_h=dev(d_h) if f & L.ROWF_STORE_H else None,
                             d_rot=dev(d_r) if f & L.ROWF_STORE_ROT else None, d_z=d_z, d_xres=d_xres,
                             d_film=d_film[:, 1024:], dfilm_ld=2048, partials=part, chunks=chunks, **common))
    gb, glg, glb, gng, gnb = (torch.zeros(512, device=DEV) for _ in range(5))
    K.row_param_reduce(part, chunks * nseq, gb, glg, glb, gng, gnb)
    # the same sums added straight into the gradients (what the training engine uses: no partials, no second launch)
    g2 = [torch.full((512,), 0.5, device=DEV) for _ in range(5)]
    d_film2 = torch.zeros(nseq, 2048, device=DEV)
    K.row_bwd(dt, K.row_args(d_xn=dev(d_xn), d_h=dev(d_h) if f & L.ROWF_STORE_H else None,
                             d_rot=dev(d_r) if f & L.ROWF_STORE_ROT else None, d_z=torch.empty_like(d_z),
                             d_xres=torch.empty_like(d_xres), d_film=d_film2[:, 1024:], dfilm_ld=2048, chunks=2, g_bias=g2[0],
                             g_ln_g=g2[1] if f & L.ROWF_LN_POST else None, g_ln_b=g2[2] if f & L.ROWF_LN_POST else None,
                             g_nln_g=g2[3] if f & L.ROWF_NEXT_LN else None, g_nln_b=g2[4] if f & L.ROWF_NEXT_LN else None,
                             **common))
    for k, (got, ref_) in enumerate(zip(g2, (gb, glg, glb, gng, gnb))):
        on = k == 0 or (k in (1, 2) and f & L.ROWF_LN_POST) or (k in (3, 4) and f & L.ROWF_NEXT_LN)
        want_k = ref_ + 0.5 if on else torch.full_like(ref_, 0.5)
        assert float((got - want_k).abs().max()) <= 1e-5 * max(1.0, float(ref_.abs().max())), k
    assert rel(d_film2, d_film) < 1e-5 or float(d_film.abs().max()) == 0.0
    gtol = max(tol, 5e-6)
    assert rel(d_z, zr.grad) < gtol
    if f & L.ROWF_RES:
        assert rel(d_xres, xr.grad) < 5e-6
    if f & L.ROWF_BIAS:
        assert rel(gb, br.grad) < 1e-5
    if f & L.ROWF_LN_POST:
        assert rel(glg, lgr.grad) < 1e-5 and rel(glb, lbr.grad) < 1e-5
    if f & L.ROWF_NEXT_LN:
        assert rel(gng, ngr.grad) < 1e-5 and rel(gnb, nbr.grad) < 1e-5
    if f & L.ROWF_FILM:
        assert rel(d_film[:, 1024:], fr.grad[:, 1024:]) < 1e-5 and float(d_film[:, :1024].abs().max()) == 0.0
    assert keepers is not None


def _images(x, Lp, T):
    """[n, H, L, 64] -> zero-padded head-major image [n, H, Lp, 64] of dtype T"""
    n, H, Lx, _ = x.shape
    img = torch.zeros(n, H, Lp, 64, dtype=T)
    img[:, :, :Lx] = x.to(T)
    return img.to(DEV)


@pytest.mark.parametrize("compute", ["f32", "bf16"])
# 450 / 300-row shapes take the operand-resident bf16 kernels (attn_res.h, attn_bwd_*_res_kernel), the others -- and every f32
# case -- the streaming ones
@pytest.mark.parametrize("Lq,Lk,p", [(120, 120, 0.1), (120, 62, 0.1), (450, 152, 0.0), (70, 130, 0.1), (450, 450, 0.1),
                                     (450, 152, 0.1), (300, 450, 0.1), (257, 33, 0.1), (600, 600, 0.1)])      # 600: two key chunks
def test_attention_train_forward_and_backward_vs_torch_autograd(compute, Lq, Lk, p):
    dt, T, tol = mode(compute)
    n, H = 2, 8
    Lpq, Lpk = K.round_up(Lq, 128), K.round_up(Lk, 128)
    g = torch.Generator().manual_seed(5)
    q, k, v = (torch.randn(n, H, Lx, 64, generator=g) for Lx in (Lq, Lk, Lk))
    do = torch.randn(n, H, Lq, 64, generator=g)
    site = 16 + 8 * 3
    thr, sc = K.drop_params(p)
    Qi, Ki, Vi, dOi = _images(q * 0.125, Lpq, T), _images(k, Lpk, T), _images(v, Lpk, T), _images(do, Lpq, T)
    O_ = torch.zeros(n * Lq, 512, device=DEV, dtype=T)
    lse = torch.zeros(n, H, Lpq, device=DEV)
    K.attention_train(dt, Qi, Ki, Vi, O_, lse, n, H, Lq, Lk, Lpq, Lpk, 512, seed_dev(), site, thr, sc)
    D = torch.float64
    qr = (Qi.cpu()[:, :, :Lq].to(D) * 8).requires_grad_(True)       # the UNSCALED projection output
    kr, vr = Ki.cpu()[:, :, :Lk].to(D).requires_grad_(True), Vi.cpu()[:, :, :Lk].to(D).requires_grad_(True)
    att = torch.softmax((qr * 0.125) @ kr.transpose(2, 3), -1)
    lse_ref = torch.logsumexp((qr * 0.125) @ kr.transpose(2, 3), -1) / math.log(2.0)
    if p > 0:
        att = att * O.dropout_keep(SEED, site, (n, H, Lq, Lk), p) / (1 - p)
    o = att @ vr
    o.backward(dOi.cpu()[:, :, :Lq].to(D))
    o_tok = o.detach().transpose(1, 2).reshape(n * Lq, 512)
    assert rel(O_, o_tok) < max(tol, 3e-6), rel(O_, o_tok)
    assert float((lse.cpu()[:, :, :Lq] - lse_ref.detach()).abs().max()) < (1e-4 if compute == "f32" else 5e-2)
    delta = torch.zeros(n, H, Lpq, device=DEV)
    dQ = torch.zeros(n * Lq, 1536, device=DEV, dtype=T)
    dKV = torch.zeros(n * Lk, 1024, device=DEV, dtype=T)
    K.attention_bwd(dt, Qi, Ki, Vi, O_, dOi, lse, delta, dQ, 1536, dKV, dKV.view(-1)[512:], 1024, n, H, Lq, Lk, Lpq, Lpk, 512,
                    0.125, seed_dev(), site, thr, sc)
    tok = lambda t, Lx: t.transpose(1, 2).reshape(n * Lx, 512)
    gtol = max(tol * 1.5, 1e-5)
    assert rel(dQ[:, :512], tok(qr.grad, Lq)) < gtol, rel(dQ[:, :512], tok(qr.grad, Lq))
    assert rel(dKV[:, :512], tok(kr.grad, Lk)) < gtol, rel(dKV[:, :512], tok(kr.grad, Lk))
    assert rel(dKV[:, 512:], tok(vr.grad, Lk)) < gtol, rel(dKV[:, 512:], tok(vr.grad, Lk))
    assert float(dQ[:, 512:].float().abs().max()) == 0.0          # nothing written outside the addressed columns
    # bitwise reproducible: no atomics in the attention backward
    dQ2, dKV2 = torch.zeros_like(dQ), torch.zeros_like(dKV)
    K.attention_bwd(dt, Qi, Ki, Vi, O_, dOi, lse, delta, dQ2, 1536, dKV2, dKV2.view(-1)[512:], 1024, n, H, Lq, Lk, Lpq, Lpk,
                    512, 0.125, seed_dev(), site, thr, sc)
    assert torch.equal(dQ, dQ2) and torch.equal(dKV, dKV2)


@pytest.mark.parametrize("Lq,Lk", [(120, 120), (450, 450), (450, 152)])      # streaming / operand-resident kernels
def test_attention_second_output_image_sharpens_the_backward_row_term(Lq, Lk):
    """tcdiff_attention_train's O_lo (round 6): O + O_lo carries 16 bits of the fp32 output, and with it the backward's
    delta_i = sum_d dO_id O_id -- the row term of dS = P (dP - delta), a cancellation -- is the exact one to ~2^-16 instead of 2^-9.
    Checked on delta itself (against float64 on the same bf16 operands) and on dQ / dK, whose error against torch autograd must not
    grow; the 8-bit image O is bit-identical with and without the second output."""
    dt, T, _ = mode("bf16")
    n, H, p = 2, 8, 0.0
    Lpq, Lpk = K.round_up(Lq, 128), K.round_up(Lk, 128)
    g = torch.Generator().manual_seed(9)
    q, k, v = (torch.randn(n, H, Lx, 64, generator=g) for Lx in (Lq, Lk, Lk))
    v = v + 1.5                                       # a common component in V: O's magnitude (and delta's) well above the row-to-row signal
    do = torch.randn(n, H, Lq, 64, generator=g)
    site = 16
    thr, sc = K.drop_params(p)
    Qi, Ki, Vi, dOi = _images(q * 0.125, Lpq, T), _images(k, Lpk, T), _images(v, Lpk, T), _images(do, Lpq, T)
    O1, O2, Olo = (torch.zeros(n * Lq, 512, device=DEV, dtype=T) for _ in range(3))
    lse = torch.zeros(n, H, Lpq, device=DEV)
    K.attention_train(dt, Qi, Ki, Vi, O1, lse, n, H, Lq, Lk, Lpq, Lpk, 512, seed_dev(), site, thr, sc)
    K.attention_train(dt, Qi, Ki, Vi, O2, lse, n, H, Lq, Lk, Lpq, Lpk, 512, seed_dev(), site, thr, sc, O_lo=Olo)
    assert torch.equal(O1, O2)
    D = torch.float64
    qr = (Qi.cpu()[:, :, :Lq].to(D) * 8).requires_grad_(True)
    kr, vr = Ki.cpu()[:, :, :Lk].to(D).requires_grad_(True), Vi.cpu()[:, :, :Lk].to(D).requires_grad_(True)
    att = torch.softmax((qr * 0.125) @ kr.transpose(2, 3), -1)
    o = att @ vr
    dO64 = dOi.cpu()[:, :, :Lq].to(D)
    o.backward(dO64)
    o_tok = o.detach().transpose(1, 2).reshape(n * Lq, 512)
    e_hi, e_both = rel(O2, o_tok), rel(O2.float() + Olo.float(), o_tok)
    delta_ref = (dO64 * o.detach()).sum(-1)
    out = {}
    for name, lo in (("O alone", None), ("O + O_lo", Olo)):
        delta = torch.zeros(n, H, Lpq, device=DEV)
        dQ = torch.zeros(n * Lq, 1536, device=DEV, dtype=T)
        dKV = torch.zeros(n * Lk, 1024, device=DEV, dtype=T)
        K.attention_bwd(dt, Qi, Ki, Vi, O2, dOi, lse, delta, dQ, 1536, dKV, dKV.view(-1)[512:], 1024, n, H, Lq, Lk, Lpq, Lpk, 512,
                        0.125, seed_dev(), site, thr, sc, O_lo=lo)
        tok = lambda t, Lx: t.transpose(1, 2).reshape(n * Lx, 512)
        out[name] = (rel(delta.cpu()[:, :, :Lq], delta_ref), rel(dQ[:, :512], tok(qr.grad, Lq)), rel(dKV[:, :512], tok(kr.grad, Lk)))
    print(f"L={Lq}x{Lk}: O vs fp64 {e_hi:.2e}, O + O_lo {e_both:.2e}; (delta, dQ, dK) vs fp64: O alone " +
          " ".join(f"{x:.2e}" for x in out["O alone"]) + " | O + O_lo " + " ".join(f"{x:.2e}" for x in out["O + O_lo"]))
    # (the float64 reference does not round P to bf16 before P V as the kernels do: that, not O's rounding, is what is left)
    assert e_both < 0.3 * e_hi
    assert out["O + O_lo"][0] < 0.25 * out["O alone"][0]
    assert out["O + O_lo"][1] <= 1.05 * out["O alone"][1] and out["O + O_lo"][2] <= 1.05 * out["O alone"][2]


@pytest.mark.parametrize("loss_type", ["l2", "l1"])
def test_loss_function_backward_vs_oracle_autograd(loss_type):
    """total(model_out) of model/diffusion.py:668-741, d total / d model_out, with contacts above 0.95 so that the
    foot-skate term and its gradient are live (random-weight outputs never reach it)."""
    from tcdiff_amd.diffusion import _LossFn
    from tcdiff_amd.fk import SMPL_OFFSETS, SMPL_PARENTS
    b, dn, S, C, T = 3, 2, 12, 151, 100
    g = torch.Generator().manual_seed(6)
    x_start = torch.randn(b, dn, S, C, generator=g) * 0.5
    out = (x_start.permute(0, 2, 1, 3).reshape(b, S * dn, C) + 0.3 * torch.randn(b, S * dn, C, generator=g)).contiguous()
    out[:, :, :4] = torch.rand(b, S * dn, 4, generator=g) * 1.2                 # some contacts > 0.95
    t = torch.tensor([5, 50, 99])
    tab = O.make_tables(T)
    od = out.to(DEV).requires_grad_(True)
    total, terms = _LossFn.apply(od, x_start.to(DEV), t.to(DEV), tab["p2_loss_weight"].to(DEV), SMPL_PARENTS, SMPL_OFFSETS,
                                 loss_type == "l1")
    (total * 1.7).backward()
    orf = out.clone().requires_grad_(True)
    # oracle p_losses with the model output injected (no denoiser evaluation)
    o_total, o_losses = O.p_losses({}, tab, x_start, None, t, torch.zeros(b, S, dn, C), None, loss_type=loss_type, model_out=orf)
    (o_total * 1.7).backward()
    got = [float(v) for v in terms]
    want = [float(v) for v in o_losses]
    assert want[3] > 0, "the foot-skate term must be live in this test"
    for a_, w_ in zip(got, want):
        assert abs(a_ - w_) < 2e-5 * max(abs(w_), 1e-3), (got, want)
    assert abs(float(total) - float(o_total)) < 2e-5 * abs(float(o_total))
    assert rel(od.grad, orf.grad) < 2e-5, rel(od.grad, orf.grad)
