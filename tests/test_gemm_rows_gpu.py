"""tcdiff_gemm_rows / tcdiff_pack_row_streams (csrc/gemm_rows.hip) through the C ABI: the device-side stream packer against the
host packer of the sampler engine, and every epilogue of the row-block GEMM against a plain fp32 torch evaluation on the same
bf16 operands (and, where the arithmetic after the product is elementwise, bit-for-bit against the separate launches)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tcdiff_amd import _lib as L  # noqa: E402
from tcdiff_amd import kernels as K  # noqa: E402
from tcdiff_amd.engine import DenoiserEngine as E  # noqa: E402

DEV = "cuda"
BF = torch.bfloat16
DT = L.DT_BF16
SEED = (1234, 5678)


def seed_dev():
    return torch.tensor(SEED, dtype=torch.int32, device=DEV)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def test_device_packer_equals_the_host_packer_in_both_orders():
    W = rnd(1536, 512, seed=1, scale=0.05)                         # fp32 master of a stacked w_qs / w_ks / w_vs
    got = K.row_streams(W)                                         # Wn = W: three phases of 16 stages
    Wb = W.to(BF).cpu()
    want = torch.cat([E._stages_n512(Wb[512 * p:512 * p + 512]) for p in range(3)], 1)
    assert got.shape == (8, 48, 2048) and torch.equal(got.cpu(), want)
    W2 = rnd(512, 1024, seed=2, scale=0.05)                        # linear2 [512, 1024]
    gotT = K.row_streams(W2, transposed=True)                      # Wn = W2^T [1024, 512]: the input gradient's operand
    W2t = W2.t().contiguous().to(BF).cpu()
    wantT = torch.cat([E._stages_n512(W2t[512 * p:512 * p + 512]) for p in range(2)], 1)
    assert torch.equal(gotT.cpu(), wantT)
    got2 = K.row_streams(W2)                                       # K = 1024: one phase of 32 stages
    assert torch.equal(got2.cpu(), E._stages_n512(W2.to(BF).cpu()))
    # a view with a leading dimension larger than its width (a row slice of a stacked parameter keeps the parent's ld)
    big = rnd(512, 1536, seed=3, scale=0.05)
    v = big[:, 512:1024]
    assert torch.equal(K.row_streams(v).cpu(), E._stages_n512(v.contiguous().to(BF).cpu()))


@pytest.mark.parametrize("mt", [4, 2, 1, 0])
@pytest.mark.parametrize("M,N,Kd", [(300, 512, 512), (917, 1024, 512), (450, 512, 1024), (64, 1536, 512)])
def test_plain_outputs_vs_torch(mt, M, N, Kd):
    A = rnd(M, Kd, seed=4).to(BF)
    W = rnd(N, Kd, seed=5, scale=0.05)
    bias = rnd(N, seed=6)
    ws = K.row_streams(W)
    want = A.float() @ W.to(BF).float().t() + bias
    ldc = N + 8
    o32 = torch.full((M, ldc), 7.0, device=DEV)
    K.gemm_rows(A, ws, M, N, Kd, mode=L.EPI_STORE_F32, bias=bias, out=o32, ldc=ldc, mt=mt)
    assert float((o32[:, :N] - want).abs().max()) < 2e-3 * float(want.abs().max())
    assert torch.all(o32[:, N:] == 7.0)                            # nothing beyond column N
    ob = torch.full((M, ldc), 7.0, device=DEV, dtype=BF)
    K.gemm_rows(A, ws, M, N, Kd, mode=L.EPI_STORE_T, bias=bias, out=ob, ldc=ldc, mt=mt)
    assert torch.equal(ob[:, :N], o32[:, :N].to(BF)) and torch.all(ob[:, N:] == 7.0)
    # without a bias, with an operand that is a column slice of a wider matrix (the input gradient of a stacked linear)
    wide = rnd(M, Kd + 512, seed=7).to(BF)
    o2 = torch.empty(M, N, device=DEV)
    K.gemm_rows(wide.view(-1)[512:], ws, M, N, Kd, lda=Kd + 512, mode=L.EPI_STORE_F32, out=o2, ldc=N, mt=mt)
    want2 = wide[:, 512:].float() @ W.to(BF).float().t()
    assert float((o2 - want2).abs().max()) < 2e-3 * float(want2.abs().max())


@pytest.mark.parametrize("mt", [4, 2, 1])
def test_qkv_heads_with_two_operands_equal_gemm_tile(mt):
    B, Lq, H = 3, 150, 8
    M, Lp = B * Lq, 256
    rot, h = rnd(M, 512, seed=8).to(BF), rnd(M, 512, seed=9).to(BF)
    W = rnd(1536, 512, seed=10, scale=0.05)
    Wb = W.to(BF)
    Q1, K1, V1 = (torch.zeros(B, H, Lp, 64, device=DEV, dtype=BF) for _ in range(3))
    K.gemm_tile(DT, rot, Wb, M, 1536, 512, A2=h, split_n=1024, mode=L.EPI_QKV_HEADS, out=Q1, out_k=K1, out_v=V1, scale_q=0.125,
                Lseq=Lq, Lp=Lp, H=H, n_q=512, n_k=512)
    Q2, K2, V2 = (torch.zeros(B, H, Lp, 64, device=DEV, dtype=BF) for _ in range(3))
    K.gemm_rows(rot, K.row_streams(W), M, 1536, 512, A2=h, split_n=1024, mode=L.EPI_QKV_HEADS, out=Q2, out_k=K2, out_v=V2,
                scale_q=0.125, Lseq=Lq, Lp=Lp, H=H, n_q=512, n_k=512, mt=mt)
    for a, b in ((Q1, Q2), (K1, K2), (V1, V2)):
        assert float((a.float() - b.float()).abs().max()) <= 2e-2 * float(a.float().abs().max())   # bf16 outputs, different summation order
        assert torch.all(b[:, :, Lq:] == 0)                        # padding rows untouched
    # exact value check of one image against fp32 torch
    want_v = (h.float() @ Wb[1024:].float().t()).view(B, Lq, H, 64).permute(0, 2, 1, 3)
    assert float((V2[:, :, :Lq].float() - want_v).abs().max()) < 1e-2 * float(want_v.abs().max())
    # with a bias (nn.MultiheadAttention's in_proj_bias, model/model.py:190-192): added before the scatter
    bias = rnd(1536, seed=21)
    Q3, K3, V3 = (torch.zeros(B, H, Lp, 64, device=DEV, dtype=BF) for _ in range(3))
    K.gemm_rows(rot, K.row_streams(W), M, 1536, 512, A2=h, split_n=1024, mode=L.EPI_QKV_HEADS, bias=bias, out=Q3, out_k=K3, out_v=V3,
                scale_q=0.125, Lseq=Lq, Lp=Lp, H=H, n_q=512, n_k=512, mt=mt)
    want_k = (rot.float() @ Wb[512:1024].float().t() + bias[512:1024]).view(B, Lq, H, 64).permute(0, 2, 1, 3)
    want_q = ((rot.float() @ Wb[:512].float().t() + bias[:512]) * 0.125).view(B, Lq, H, 64).permute(0, 2, 1, 3)
    assert float((K3[:, :, :Lq].float() - want_k).abs().max()) < 1e-2 * float(want_k.abs().max())
    assert float((Q3[:, :, :Lq].float() - want_q).abs().max()) < 1e-2 * float(want_q.abs().max())
    # a single image (dO of the attention backward, cross-attention's Q): n_q = 512, nothing else
    dO = torch.zeros(B, H, Lp, 64, device=DEV, dtype=BF)
    K.gemm_rows(h, K.row_streams(W[1024:]), M, 512, 512, mode=L.EPI_QKV_HEADS, out=dO, scale_q=1.0, Lseq=Lq, Lp=Lp, H=H, n_q=512,
                n_k=0, mt=mt)
    assert torch.equal(dO, V2)


@pytest.mark.parametrize("mt", [4, 1])
@pytest.mark.parametrize("act,p", [(L.ACT_GELU, 0.1), (L.ACT_RELU, 0.0)])
def test_activation_epilogues_equal_the_separate_launches(mt, act, p):
    M, N, Kd = 333, 1024, 512
    A = rnd(M, Kd, seed=11).to(BF)
    W = rnd(N, Kd, seed=12, scale=0.05)
    bias = rnd(N, seed=13)
    ws = K.row_streams(W)
    thr, sc = K.drop_params(p)
    site = 22
    a1 = torch.empty(M, N, device=DEV, dtype=BF)
    K.gemm_rows(A, ws, M, N, Kd, bias=bias, out=a1, ldc=N, mt=mt)
    y1 = torch.empty(M, N, device=DEV, dtype=BF)
    K.act_drop(DT, a1, N, y1, N, M, N, act, seed_dev(), site, thr, sc)
    a2, y2 = torch.empty(M, N, device=DEV, dtype=BF), torch.empty(M, N, device=DEV, dtype=BF)
    K.gemm_rows(A, ws, M, N, Kd, bias=bias, out=a2, ldc=N, out2=y2, ldc2=N, act2=act, seed=seed_dev(), site=site, thr=thr,
                drop_scale=sc, mt=mt)
    assert torch.equal(a1, a2) and torch.equal(y1, y2)
    if p > 0:
        assert 0.05 < float((y2 == 0).float().mean()) < 0.6
    # the same dropout bits as gemm_tile's fused epilogue (element index m * N + n)
    a3, y3 = torch.empty(M, N, device=DEV, dtype=BF), torch.empty(M, N, device=DEV, dtype=BF)
    K.gemm_tile(DT, A, W.to(BF), M, N, Kd, bias=bias, out=a3, ldc=N, out2=y3, ldc2=N, act2=act, seed=seed_dev(), site=site,
                thr=thr, drop_scale=sc)
    assert torch.equal(y2 == 0, y3 == 0) or p == 0
    # backward: dA = (dY W2) through the activation; W2 [512, 1024] is linear2's weight, its input gradient reads it transposed
    dY = rnd(M, 512, seed=14).to(BF)
    W2 = rnd(512, N, seed=15, scale=0.05)
    wsT = K.row_streams(W2, transposed=True)
    d1, da1 = torch.empty(M, N, device=DEV, dtype=BF), torch.empty(M, N, device=DEV, dtype=BF)
    K.gemm_rows(dY, wsT, M, N, 512, out=d1, ldc=N, mt=mt)
    K.act_drop_bwd(DT, a1, N, d1, N, da1, M, N, act, seed_dev(), site, thr, sc)
    da2 = torch.empty(M, N, device=DEV, dtype=BF)
    K.gemm_rows(dY, wsT, M, N, 512, out=da2, ldc=N, act_src=a1, ld_src=N, act2=act, seed=seed_dev(), site=site, thr=thr,
                drop_scale=sc, mt=mt)
    assert torch.equal(da1, da2)
    want = dY.float() @ W2.to(BF).float()
    assert float((d1.float() - want).abs().max()) < 1.5e-2 * float(want.abs().max())


def test_refused_shapes_and_arguments():
    A = rnd(64, 768, seed=1).to(BF)
    ws = torch.zeros(8, 16, 2048, device=DEV, dtype=BF)
    out = torch.empty(64, 512, device=DEV)
    assert not K.gemm_rows_ok(DT, 512, 768) and not K.gemm_rows_ok(L.DT_F32, 512, 512) and K.gemm_rows_ok(DT, 1536, 1024)
    with pytest.raises(L.TcdiffError):                             # K = 768
        K.gemm_rows(A, torch.zeros(8, 24, 2048, device=DEV, dtype=BF), 64, 512, 768, mode=L.EPI_STORE_F32, out=out, ldc=512)
    with pytest.raises(L.TcdiffError):                             # stream of the wrong length
        K.gemm_rows(A, ws, 64, 1024, 512, mode=L.EPI_STORE_F32, out=out, ldc=1024)
    with pytest.raises(L.TcdiffError):                             # output narrower than N
        K.gemm_rows(A, ws, 64, 512, 512, lda=768, mode=L.EPI_STORE_F32, out=out, ldc=256)
    with pytest.raises(L.TcdiffError):                             # two images in one 512-column phase
        K.gemm_rows(A, ws, 64, 512, 512, lda=768, mode=L.EPI_QKV_HEADS, out=out, out_k=out, Lseq=64, Lp=128, H=8, n_q=256, n_k=256)


def test_full_size_products_agree_with_gemm_tile_and_are_linear_in_the_rows():
    """The training step's shape (28 800 token rows = 32 clips of 3 x 150): the row-block GEMM against the tile GEMM on the same
    operands (same bf16 products, fp32 accumulation in another order), and a size-independent property -- output rows depend on
    their own input row only: permuting the rows of A permutes the rows of C bit for bit, and a block of rows computed alone
    (other block size, other workgroup) equals the same rows of the full product."""
    M, N, Kd = 28800, 1024, 512
    A = rnd(M, Kd, seed=31).to(BF)
    W = rnd(N, Kd, seed=32, scale=0.05)
    bias = rnd(N, seed=33)
    ws = K.row_streams(W)
    c_rows, c_tile = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    K.gemm_rows(A, ws, M, N, Kd, mode=L.EPI_STORE_F32, bias=bias, out=c_rows, ldc=N)
    K.gemm_tile(DT, A, W.to(BF), M, N, Kd, mode=L.EPI_STORE_F32, bias=bias, out=c_tile, ldc=N)
    assert float((c_rows - c_tile).abs().max()) <= 1e-5 * float(c_tile.abs().max())       # fp32 sums of identical products
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(5)).to(DEV)
    c_perm = torch.empty(M, N, device=DEV)
    K.gemm_rows(A[perm].contiguous(), ws, M, N, Kd, mode=L.EPI_STORE_F32, bias=bias, out=c_perm, ldc=N)
    assert torch.equal(c_perm, c_rows[perm])
    sub = torch.empty(48, N, device=DEV)
    K.gemm_rows(A[1000:1048].contiguous(), ws, 48, N, Kd, mode=L.EPI_STORE_F32, bias=bias, out=sub, ldc=N, mt=1)
    assert torch.equal(sub, c_rows[1000:1048])
