"""The small-job form of the decoder layer (csrc/chain_split.hip, tcdiff_chain_split): four workgroups per 16-row block, four
launches per layer -- self-attention of two heads per workgroup with the keys dealt to four waves, fc / linear2 split over the
contraction, w_qs / linear1 / Q, K, V split over the output columns -- against the ONE fused launch it replaces on small jobs
(TC_CHAIN_FULL at 16-row blocks, which tests/test_chain_gpu.py, test_chain_selfatt_gpu.py and the parity goldens hold to the
reference).  Same weights stream, same fragment images in and out; the arithmetic differs in fp32 summation order only (four
partial sums per product, four partial softmaxes per head)."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from tcdiff_amd import _lib as L  # noqa: E402
from tcdiff_amd import kernels as K  # noqa: E402
from tcdiff_amd.model import DanceDecoder  # noqa: E402
from oracle import tcdiff_oracle as O  # noqa: E402
from tcdiff_amd.engine import DenoiserEngine as E  # noqa: E402
from test_chain_selfatt_gpu import DEV, Layer, bf, kf_index, rnd, unpack_kv, unpack_q, vf_index  # noqa: E402


def run_layers(Lq, nseq, split, qk_gain=1.0, Lk=62, stamps=None):
    H = 8
    M = nseq * Lq
    Lp = K.round_up(Lq, 128)
    Lpc, nkt = K.round_up(Lk, 128), (Lk + 31) // 32
    n_kv = nseq
    l0, l1, l2 = Layer(100, nseq, False, qk_gain), Layer(200, nseq, False), Layer(300, nseq, True)
    Oa = rnd(M, 512, seed=51, scale=0.5).to(bf)
    xres = rnd(M, 512, seed=94)
    rope = torch.empty(Lq, 512, device=DEV)
    K.rope_table((1.0 / (10000 ** (torch.arange(0, 512, 2).float() / 512))).to(DEV), rope, Lq)
    rope = K.to_cb(rope)
    z = lambda *s, dtype=bf: torch.zeros(*s, device=DEV, dtype=dtype)
    Kc, Vc = z(n_kv, H, Lpc, 64), z(n_kv, H, Lpc, 64)
    Kc[:, :, :Lk] = rnd(n_kv, H, Lk, 64, seed=95).to(bf)
    Vc[:, :, :Lk] = rnd(n_kv, H, Lk, 64, seed=96).to(bf)
    Kf, Vf = z(n_kv, H, nkt * 2048), z(n_kv, H, nkt * 2048)
    K.pack_kv_frags(Kc, Vc, Kf, Vf, n_kv, H, Lpc, nkt, 0, Lk)
    xatt = dict(kf=Kf, vf=Vf, n_shared=1, nkt=nkt, Lk=Lk, rope=rope, Lp=Lp, mt=1, seq_blocks=True)
    nbs, skt = (Lq + 15) // 16, (Lq + 31) // 32
    X = [K.to_cb(xres), torch.zeros(64, M, 8, device=DEV)]
    P = [torch.zeros(nseq * nbs, 4, 16, 512, device=DEV) for _ in range(2)]
    qf = [z(nseq * nbs, 8, 4, 2, 64, 8) for _ in range(2)]
    skf = [torch.full((nseq, H, skt * 2048), float("nan"), device=DEV, dtype=bf) for _ in range(2)]
    svf = [torch.full((nseq, H, skt * 2048), float("nan"), device=DEV, dtype=bf) for _ in range(2)]
    h = torch.zeros(M, 512, device=DEV, dtype=bf)
    outs = {}
    cur = 0
    for li, lay in enumerate((l0, l1, l2)):
        last = li == 2
        kw = dict(lay.kw, **xatt)
        if li > 0:
            kw.update(sa_q=qf[(li - 1) & 1], sa_kf=skf[(li - 1) & 1], sa_vf=svf[(li - 1) & 1], sa_nkt=skt)
        if last:
            kw.update(h_out=h)
        else:
            kw.update(qf_out=qf[li & 1], kf_out=skf[li & 1], vf_out=svf[li & 1], out_nkt=skt)
            if stamps is not None:            # (tools/split_stamps.py, -DCS_STAMP builds: the buffer the rows of a last layer would go to)
                kw.update(h_out=stamps)
        if not split:
            K.chain(lay.mode, M, Lq, Oa, lay.ws, xres=X[cur], xout=X[cur], **kw)
        else:
            o1 = cur ^ 1
            if split == "merged":         # parts 1 + 2 as one launch
                K.chain(lay.mode, M, Lq, Oa, lay.ws, split_part=12, p_out=P[1], xres=X[cur], xout=X[o1], **kw)
            else:
                K.chain(lay.mode, M, Lq, Oa, lay.ws, split_part=1, p_out=P[0], xres=X[cur], xout=X[o1], **kw)
                K.chain(lay.mode, M, Lq, Oa, lay.ws, split_part=2, p_in=P[0], p_out=P[1], xres=X[cur], xout=X[o1], **kw)
            K.chain(lay.mode, M, Lq, Oa, lay.ws, split_part=3, p_in=P[1], p_out=P[0], xres=X[o1], xout=X[cur], **kw)
            K.chain(lay.mode, M, Lq, Oa, lay.ws, split_part=4, p_in=P[0], xres=X[cur], xout=X[o1], **kw)
            cur = o1
        if not last:
            outs[f"x{li}"] = K.from_cb(X[cur], M).clone()
            outs[f"q{li}"], outs[f"k{li}"], outs[f"v{li}"] = qf[li & 1].clone(), skf[li & 1].clone(), svf[li & 1].clone()
    torch.cuda.synchronize()
    outs["h"] = h.float().clone()
    return outs


# 450 = 28 x 16 + 2 (a last block of 2 rows), 120 / 150: the small configurations (C1: 2 x 60; 3 x 150 = 450 per clip); 62 / 152 keys:
# the cross-attention memories of those configurations (2 / 5 key tiles: heads whose four waves do not all get a tile)
@pytest.mark.parametrize("form", [True, "merged"])
@pytest.mark.parametrize("Lq,nseq,Lk,gain", [(120, 2, 62, 1.0), (450, 2, 152, 1.0), (150, 3, 62, 1.0), (137, 2, 152, 1.0), (450, 1, 152, 4.0)])
def test_split_layers_equal_the_fused_launch(Lq, nseq, Lk, gain, form):
    """Three consecutive layers (the first reads attention-output rows, the others compute their self-attention from the fragments
    the layer before left; the last is the folded *_LAST form) through tcdiff_chain_split's four parts and through the fused
    launch: residual stream, Q / K / V fragment images and the final rows.  form "merged": parts 1 and 2 as one launch (part 12:
    self-attention of all eight heads and the whole fc in every member)."""
    a, b = run_layers(Lq, nseq, False, gain, Lk), run_layers(Lq, nseq, form, gain, Lk)
    bad = []
    for k in a:
        x, y = a[k].float(), b[k].float()
        ok = ~(x.isnan() & y.isnan())            # fragment slots nobody owns stay as the poison they were allocated with in both
        assert bool((x.isnan() == y.isnan()).all()), k
        assert torch.isfinite(y[ok]).all(), k
        d = (x[ok] - y[ok]).abs()
        top = float(x[ok].abs().max())
        rel_mean, rel_max = float(d.mean()) / top, float(d.max()) / top
        print(f"L={Lq} n={nseq} Lk={Lk} gain={gain} {k}: max diff {float(d.max()):.2e} mean {float(d.mean()):.2e} of max |.| {top:.2f}"
              f" -> {rel_max:.1e} / {rel_mean:.1e}")
        # The two forms round the same fp32 values to bf16 operands at every stage, in a different summation order: isolated one-ulp
        # flips (2^-8 of the value), carried and amplified by the layers behind.  Measured against the largest magnitude of the output:
        # the first layer's outputs sit at 3e-3..7e-3 (max) / 1.5e-4..6e-4 (mean) -- one bf16 ulp of the top binade is 7.8e-3.
        first = k.endswith("0")
        if rel_mean > (1.5e-3 if first else 4e-3) or rel_max > (2e-2 if first else 6e-2):
            bad.append((k, rel_max, rel_mean))
    assert not bad, bad


@pytest.mark.parametrize("Lq,nseq", [(120, 2), (450, 1), (137, 3)])
def test_front_part_writes_the_fragment_images_of_layer_0(Lq, nseq):
    """tcdiff_chain_split part 0 (TC_CHAIN_FRONT stream, stages 32 / 48 / 64): norm1 + rotary of row-major fp32 token rows -> Q, K;
    norm1 -> V (model/model.py:326,374-383,78-80 for layer 0), as the fragment images part 1 reads, against a float64 torch evaluation
    with the kernel's rounding points (bf16 activations into the products, bf16 images out).  Rows of a last partial block (450 = 28 x
    16 + 2, 137 = 8 x 16 + 9) and the slots nobody owns (NaN-poisoned images: a key slot past the sequence must come out finite)."""
    H, M = 8, nseq * Lq
    Wf3 = rnd(512, 1024, seed=1, scale=1024 ** -0.5).to(bf)                    # (the front stream's first 32 stages: unused by part 0)
    Wqkv = rnd(1536, 512, seed=2, scale=512 ** -0.5).to(bf)
    ws = torch.cat([E._stages_n512(Wf3)] + [E._stages_n512(Wqkv[i * 512:(i + 1) * 512]) for i in range(3)], 1).contiguous()
    assert ws.shape[1] == 80
    x = rnd(M, 512, seed=3)
    g, b = 1.0 + 0.1 * rnd(512, seed=4), 0.1 * rnd(512, seed=5)
    rope = torch.empty(Lq, 512, device=DEV)
    K.rope_table((1.0 / (10000 ** (torch.arange(0, 512, 2).float() / 512))).to(DEV), rope, Lq)
    nbs, skt = (Lq + 15) // 16, (Lq + 31) // 32
    qf = torch.zeros(nseq * nbs, 8, 4, 2, 64, 8, device=DEV, dtype=bf)
    kf = torch.full((nseq, H, skt * 2048), float("nan"), device=DEV, dtype=bf)
    vf = torch.full((nseq, H, skt * 2048), float("nan"), device=DEV, dtype=bf)
    K.chain(L.CHAIN_FRONT, M, Lq, None, ws, split_part=0, xres=x, nn_g=g, nn_b=b, nn_eps=1e-5, rope=K.to_cb(rope), qf_out=qf, kf_out=kf,
            vf_out=vf, out_nkt=skt, scale_q=0.125, H=H)
    torch.cuda.synchronize()
    # reference: LayerNorm in float64, rotation by the same table, bf16 operands
    xd = x.double()
    h = (xd - xd.mean(1, keepdim=True)) / torch.sqrt(xd.var(1, unbiased=False, keepdim=True) + 1e-5) * g.double() + b.double()
    rp = rope.double().repeat(nseq, 1).reshape(M, 256, 2)                    # (cos, sin) per pair
    hp = h.reshape(M, 256, 2)
    rot = torch.stack([hp[..., 0] * rp[..., 0] - hp[..., 1] * rp[..., 1], hp[..., 1] * rp[..., 0] + hp[..., 0] * rp[..., 1]], -1).reshape(M, 512)
    hb, rb = h.float().to(bf).double(), rot.float().to(bf).double()
    W = Wqkv.double()
    heads = lambda t: t.reshape(nseq, Lq, H, 64).permute(0, 2, 1, 3)
    q_ref = heads(rb @ W[:512].T) * (0.125 * 1.4426950408889634)              # the in-launch softmax works in the exp2 domain
    k_ref, v_ref = heads(rb @ W[512:1024].T), heads(hb @ W[1024:].T)
    q = unpack_q(qf, nseq, Lq, 16).double()
    k, v = unpack_kv(kf, kf_index, Lq).double(), unpack_kv(vf, vf_index, Lq).double()
    for nm, got, ref in (("q", q, q_ref), ("k", k, k_ref), ("v", v, v_ref)):
        d = (got - ref).abs()
        print(f"part 0, {nseq} x {Lq}: {nm} max-abs {float(d.max()):.2e} mean {float(d.mean()):.2e} (max |.| {float(ref.abs().max()):.2f})")
        assert torch.isfinite(got).all() and float(d.max()) < 2.5e-2 * max(1.0, float(ref.abs().max())) and float(d.mean()) < 2.5e-3, nm
    # every V^T slot of the last key tile is finite (P = 0 times NaN would poison the in-launch attention)
    assert torch.isfinite(vf.float()).all()


def test_launcher_refuses_in_place_and_missing_buffers():
    lay = Layer(7, 1, False)
    z = lambda *s, dtype=bf: torch.zeros(*s, device=DEV, dtype=dtype)
    x = torch.zeros(64, 64, 8, device=DEV)
    rope = torch.zeros(64, 64, 8, device=DEV)
    P = torch.zeros(4, 4, 16, 512, device=DEV)
    kf = z(1, 8, 2 * 2048)
    kw = dict(lay.kw, kf=kf, vf=kf, n_shared=1, nkt=2, Lk=62, rope=rope, Lp=128, mt=1, seq_blocks=True, qf_out=z(4, 8, 4, 2, 64, 8),
              kf_out=z(1, 8, 2 * 2048), vf_out=z(1, 8, 2 * 2048), out_nkt=2)
    with pytest.raises(L.TcdiffError):            # parts 2-4 are not in place
        K.chain(lay.mode, 64, 64, z(64, 512), lay.ws, split_part=2, p_in=P, p_out=P, xres=x, xout=x, **kw)
    with pytest.raises(L.TcdiffError):            # a part that reads partial sums needs them
        K.chain(lay.mode, 64, 64, z(64, 512), lay.ws, split_part=3, p_out=P, xres=x, xout=torch.zeros_like(x), **kw)


# (1 x 150: one dancer; 5 x 40: 200-token sequences, 12.5 blocks; 2 x 37: 74 tokens, a 10-row last block, 37 frames in the front products)
@pytest.mark.parametrize("dn,S,B", [(2, 60, 1), (3, 150, 1), (2, 60, 3), (1, 150, 1), (5, 40, 1), (2, 37, 2)])
def test_small_job_network_matches_the_fused_layers(dn, S, B, monkeypatch):
    """The whole denoiser on a job small enough for the four-workgroups-per-block layers (both CFG branches stacked -- layer 0's
    Q / K / V shared by the branches -- and a plain forward): TCDIFF_SPLIT=0 (one fused launch per layer, TC_CHAIN_FRONT, the
    stand-alone layer-0 attention; held to the reference by the parity goldens) against the small-job form with the TC_CHAIN_FRONT
    launch (TCDIFF_SPLIT_FRONT=0) and with the fragment front (the default: fusion projection as small products, part 0, layer 0's
    self-attention in part 1).  Same bound as the in-launch-attention network test: independent bf16 rounding draws behind layer 2."""
    Lq = dn * S
    cond = torch.stack([O.synth_cond(c, S) for c in range(B)]).to(DEV)
    x = torch.stack([O.synth_xT(c, Lq) for c in range(B)]).to(DEV)
    m = DanceDecoder(nfeats=151, seq_len=S, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                     cond_feature_dim=438, activation=F.gelu, required_dancer_num=dn, compute_dtype="bf16")
    m.load_state_dict(O.synth_state_dict(dn=dn, seq_len=S))
    m.to(DEV).eval()
    outs = {}
    # (the small-job workspaces are planned with the engine: the default form first, the switches afterwards)
    for name, env in (("split+front", dict(TCDIFF_SPLIT_MERGE="0")), ("merged+front", dict(TCDIFF_SPLIT_MERGE="1")),
                      ("split", dict(TCDIFF_SPLIT_FRONT="0", TCDIFF_SPLIT_MERGE="0")), ("merged", dict(TCDIFF_SPLIT_FRONT="0", TCDIFF_SPLIT_MERGE="1")),
                      ("fused", dict(TCDIFF_SPLIT="0"))):
        for k in ("TCDIFF_SPLIT", "TCDIFF_SPLIT_FRONT", "TCDIFF_SPLIT_MERGE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        tt = torch.full((B,), 640, dtype=torch.long, device=DEV)
        g = m.guided_forward(x, cond, tt, 2.0)
        with torch.no_grad():
            f = m(x, cond, torch.arange(B, device=DEV) * 37 + 5, cond_drop_prob=0.0)
        eng = list(m._engines.values())
        assert all(e._split_job(2 * B) == (name != "fused") for e in eng if "xb" in e.b), name
        assert any(getattr(e, "_frag_front", False) for e in eng) == name.endswith("+front"), name
        outs[name] = (g.clone(), f.clone())
    for name in ("split", "split+front", "merged", "merged+front"):
        for nm, a, b in zip(("guided", "forward"), outs["fused"], outs[name]):
            d, mean = float((a - b).abs().max()), float((a - b).abs().mean())
            print(f"{name} vs fused ({dn}x{S}, B={B}, {nm}): max-abs {d:.2e}, mean-abs {mean:.2e}, |out| max {float(a.abs().max()):.2f}")
            assert torch.isfinite(b).all()
            assert d < 3e-2 and mean < 4e-3, (name, nm, d, mean)
