"""Training-side rows on CPU: the oracle restatements against (a) golden vectors from the REAL reference
(tests/golden/make_golden_train.py: p_losses reconstruction + velocity terms, Adan steps) and (b) an independent
implementation (scipy.spatial.transform) for the rotation / forward-kinematics arithmetic that lives in the absent
pytorch3d ("parity unpinned": oracle/tcdiff_oracle.py header of that block)."""
import os

import numpy as np
import torch
from scipy.spatial.transform import Rotation as R

from oracle import tcdiff_oracle as O


def test_p_losses_recon_and_velocity_terms_vs_reference_golden(golden_dir):
    ref = np.load(os.path.join(golden_dir, "c1_p_losses.npz"))
    dn, S, T, b = 2, 60, 100, 3
    sd = O.synth_state_dict(dn=dn, seq_len=S)
    x_start = torch.stack([O.synth_motion(c, dn * S).reshape(S, dn, 151).permute(1, 0, 2) for c in range(b)])
    cond = torch.stack([O.synth_cond(c, S) for c in range(b)])
    noise = torch.stack([O.synth_xT(10 + c, dn * S).reshape(S, dn, 151) for c in range(b)])
    t, keep = torch.from_numpy(ref["t"]), torch.from_numpy(ref["keep"])
    tab = O.make_tables(T)
    # the noised input the reference fed to its model (trajectory channels restored, model/diffusion.py:645-651)
    xs = x_start.permute(0, 2, 1, 3)
    xn = O.q_sample(tab, xs, t, noise).clone()
    xn[:, :, :, [4, 5]] = xs[:, :, :, [4, 5]]
    assert np.array_equal(xn.reshape(b, S * dn, 151).numpy(), ref["x_noisy"])
    with torch.no_grad():
        total, losses = O.p_losses(sd, tab, x_start, cond, t, noise, keep, with_fk=False)
    assert abs(float(losses[0]) - float(ref["recon"])) < 1e-6 * float(ref["recon"])
    assert abs(float(losses[1]) - float(ref["velocity"])) < 1e-6 * float(ref["velocity"])
    with torch.no_grad():
        out = O.decoder_forward(sd, xn.reshape(b, S * dn, 151), cond, t, keep_mask=keep)
    assert float((out - torch.from_numpy(ref["model_out"])).abs().max()) < 1e-4


def test_adan_restatement_is_bit_exact_vs_reference_golden(golden_dir):
    ref = np.load(os.path.join(golden_dir, "adan_steps.npz"))
    for i in range(3):
        p = ref[f"p{i}_init"].copy()
        st = dict(step=0, m=np.zeros_like(p), v=np.zeros_like(p), n=np.zeros_like(p), prev_grad=np.zeros_like(p))
        for step in range(int(ref["n_steps"])):
            p = O.adan_step(p, ref[f"g{i}_step{step}"], st, lr=5e-5, weight_decay=0.02)
            assert np.array_equal(p, ref[f"p{i}_step{step}"]), (i, step)
        for k in ("m", "v", "n", "prev_grad"):
            assert np.array_equal(st[k], ref[f"{k}{i}_final"]), (i, k)
    # first-step quirk (model/adan.py:71): m, v, n untouched, only weight decay acts
    p0, g0 = ref["p1_init"], ref["g1_step0"]
    st = dict(step=0, m=np.zeros_like(p0), v=np.zeros_like(p0), n=np.zeros_like(p0), prev_grad=np.zeros_like(p0))
    p1 = O.adan_step(p0.copy(), g0, st, lr=5e-5, weight_decay=0.02)
    assert np.array_equal(p1, p0 / np.float32(1 + 0.02 * 5e-5)) and not st["m"].any() and not st["n"].any()


def _rand_rot6d(n, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 6, generator=g, dtype=torch.float64)


def test_rotation_conversions_against_scipy():
    d6 = _rand_rot6d(500, 1)
    m = O.rotation_6d_to_matrix(d6)
    eye = torch.eye(3, dtype=torch.float64)
    assert float((m @ m.transpose(-1, -2) - eye).abs().max()) < 1e-12 and float((torch.linalg.det(m) - 1).abs().max()) < 1e-12
    # rows b1, b2 are the Gram-Schmidt images of the two 3-vectors
    b1 = d6[:, :3] / d6[:, :3].norm(dim=-1, keepdim=True)
    assert float((m[:, 0] - b1).abs().max()) < 1e-12
    aa = O.matrix_to_axis_angle(m)
    # same ROTATION as scipy's (pytorch3d's quaternion may have a negative real part, i.e. an angle in (pi, 2 pi):
    # the rotation vector then differs from scipy's canonical one by a full turn about the axis)
    assert np.abs(R.from_rotvec(aa.numpy()).as_matrix() - m.numpy()).max() < 1e-9
    want = R.from_matrix(m.numpy()).as_rotvec()
    canon = aa.norm(dim=-1) <= np.pi
    assert int(canon.sum()) > 100 and np.abs(aa.numpy()[canon.numpy()] - want[canon.numpy()]).max() < 1e-9
    q = O.axis_angle_to_quaternion(aa)
    sq = R.from_rotvec(aa.numpy()).as_quat()          # scipy: x y z w
    sq = np.concatenate([sq[:, 3:], sq[:, :3]], 1)
    sq = sq * np.sign(sq[:, :1] + 1e-300)
    assert np.abs(q.numpy() * np.sign(q.numpy()[:, :1] + 1e-300) - sq).max() < 1e-9
    pts = torch.randn(500, 3, dtype=torch.float64, generator=torch.Generator().manual_seed(2))
    assert np.abs(O.quaternion_apply(q, pts).numpy() - R.from_rotvec(aa.numpy()).apply(pts.numpy())).max() < 1e-9
    q2 = O.axis_angle_to_quaternion(O.matrix_to_axis_angle(O.rotation_6d_to_matrix(_rand_rot6d(500, 3))))
    prod = O.quaternion_multiply(q, q2).numpy()
    sp = (R.from_quat(np.concatenate([q.numpy()[:, 1:], q.numpy()[:, :1]], 1)) *
          R.from_quat(np.concatenate([q2.numpy()[:, 1:], q2.numpy()[:, :1]], 1))).as_quat()
    sp = np.concatenate([sp[:, 3:], sp[:, :3]], 1)
    sp = sp * np.sign(sp[:, :1] + 1e-300)
    assert np.abs(prod - sp).max() < 1e-9 and (prod[:, 0] >= 0).all()
    # small-angle branch (|angle| < 1e-6): Taylor value, finite
    tiny = torch.tensor([[1e-9, -2e-9, 5e-10]], dtype=torch.float64)
    qt = O.axis_angle_to_quaternion(tiny)
    assert torch.isfinite(qt).all() and float((O.quaternion_to_axis_angle(qt) - tiny).abs().max()) < 1e-15


def test_smpl_forward_kinematics_against_scipy_chain():
    g = torch.Generator().manual_seed(5)
    aa = torch.randn(2, 7, 24, 3, generator=g, dtype=torch.float64) * 0.7
    root = torch.randn(2, 7, 3, generator=g, dtype=torch.float64)
    got = O.smpl_fk(aa, root).numpy()
    off = np.array(O.SMPL_OFFSETS)
    for n in range(2):
        for l in range(7):
            rw, pw = [None] * 24, [None] * 24
            for j, p in enumerate(O.SMPL_PARENTS):
                if p == -1:
                    rw[j], pw[j] = R.from_rotvec(aa[n, l, 0].numpy()), root[n, l].numpy()
                else:
                    pw[j] = rw[p].apply(off[j]) + pw[p]
                    rw[j] = rw[p] * R.from_rotvec(aa[n, l, j].numpy())
            assert np.abs(got[n, l] - np.stack(pw)).max() < 1e-9


def test_p_losses_fk_and_foot_terms_are_finite_and_consistent():
    dn, S, T, b = 2, 60, 100, 2
    sd = O.synth_state_dict(dn=dn, seq_len=S)
    x_start = torch.stack([O.synth_motion(c, dn * S).reshape(S, dn, 151).permute(1, 0, 2) for c in range(b)])
    cond = torch.stack([O.synth_cond(c, S) for c in range(b)])
    noise = torch.stack([O.synth_xT(10 + c, dn * S).reshape(S, dn, 151) for c in range(b)])
    with torch.no_grad():
        total, losses = O.p_losses(sd, O.make_tables(T), x_start, cond, torch.tensor([30, 3]), noise,
                                   torch.tensor([True, True]))
    assert all(bool(torch.isfinite(l)) for l in losses) and abs(float(total) - sum(float(l) for l in losses)) < 1e-6
    assert float(losses[2]) > 0


# ---- hand-written reverse mode of the FK / 6-D conversion (csrc/fk_math.h), compiled for the HOST -----------------------
def _fk_host(tmp_path):
    import ctypes as C
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / "fk_host.so")
    subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-I" + os.path.join(root, "tcdiff_amd", "csrc"), "-o", so,
                           os.path.join(root, "tests", "host", "fk_host.cpp")])
    return C.CDLL(so)


def test_fk_math_reverse_mode_on_the_host(tmp_path):
    """The SAME source the HIP kernels compile (tcdiff_amd/csrc/fk_math.h) built with g++: forward values and the
    hand-derived adjoints of ax_from_6v and of the SMPL chain against torch autograd through the oracle's restatement."""
    import ctypes as C
    lib = _fk_host(tmp_path)
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    g = torch.Generator().manual_seed(0)
    n = 3000
    d6 = torch.randn(n, 6, generator=g)
    cot = torch.randn(n, 3, generator=g)
    d6r = d6.clone().requires_grad_(True)
    aa = O.ax_from_6v(d6r)
    (aa * cot).sum().backward()
    out, g6 = np.zeros((n, 3), np.float32), np.zeros((n, 6), np.float32)
    d6n, cn = d6.numpy().copy(), cot.numpy().copy()
    lib.host_ax_from_6v(fp(d6n), C.c_long(n), fp(out))
    lib.host_ax_from_6v_bwd(fp(d6n), fp(cn), C.c_long(n), fp(g6))
    assert np.abs(out - aa.detach().numpy()).max() < 1e-5
    ref = d6r.grad.numpy()
    assert np.isfinite(ref).all() and np.isfinite(g6).all()
    assert np.linalg.norm(g6 - ref) / np.linalg.norm(ref) < 1e-5
    N = 400
    rot, root, gj = torch.randn(1, N, 24, 3, generator=g) * 0.8, torch.randn(1, N, 3, generator=g), torch.randn(1, N, 24, 3, generator=g)
    rr, tr = rot.clone().requires_grad_(True), root.clone().requires_grad_(True)
    j = O.smpl_fk(rr, tr)
    (j * gj).sum().backward()
    par = (C.c_int * 24)(*O.SMPL_PARENTS)
    off = np.array(O.SMPL_OFFSETS, np.float32)
    jo, gaa, gr = np.zeros((N, 24, 3), np.float32), np.zeros((N, 24, 3), np.float32), np.zeros((N, 3), np.float32)
    an, rn, gn = rot[0].numpy().copy(), root[0].numpy().copy(), gj[0].numpy().copy()
    lib.host_fk(fp(an), fp(rn), C.c_long(N), par, fp(off), fp(jo))
    lib.host_fk_bwd(fp(an), fp(gn), C.c_long(N), par, fp(off), fp(gaa), fp(gr))
    assert np.abs(jo - j[0].detach().numpy()).max() < 5e-6
    assert np.abs(gaa - rr.grad[0].numpy()).max() < 2e-5 * float(rr.grad.abs().max())
    assert np.abs(gr - tr.grad[0].numpy()).max() < 1e-5


def test_dropout_hash_of_the_oracle_is_a_fair_independent_mask():
    """oracle.dropout_keep == tcdiff_amd/csrc/train_common.h (held equal on the GPU by tests/test_train_kernels_gpu.py):
    keep probability 1 - p, sites and seeds give unrelated masks, the mask is a pure function of its arguments."""
    a = O.dropout_keep((1234, 5678), 17, (64, 512), 0.1)
    b = O.dropout_keep((1234, 5678), 18, (64, 512), 0.1)
    c = O.dropout_keep((1235, 5678), 17, (64, 512), 0.1)
    assert abs(float(a.float().mean()) - 0.9) < 0.01 and abs(float(b.float().mean()) - 0.9) < 0.01
    for other in (b, c):
        agree = float((a == other).float().mean())
        assert abs(agree - (0.81 + 0.01)) < 0.01           # independent masks agree with probability 0.9^2 + 0.1^2
    assert torch.equal(a, O.dropout_keep((1234, 5678), 17, (64, 512), 0.1))
    assert bool(O.dropout_keep((1, 2), 3, (10,), 0.0).all())
    # train-mode oracle: the plan scales kept values by 1 / (1 - p)
    x = torch.ones(64, 512)
    y = O.DropPlan((1234, 5678), 0.1)(x, 17)
    assert torch.equal(y != 0, a) and abs(float(y.max()) - 1 / 0.9) < 1e-6


def test_operand_rounding_mode_of_the_oracle_rounds_and_restores():
    """oracle.operand_rounding (the bf16 rounding-point emulation the GPU training tests hold the bf16 step to): inside the context
    the denoiser's output moves by bf16-operand noise, gradients flow, and on exit the exact fp32 arithmetic is back"""
    import torch
    from oracle import tcdiff_oracle as O
    dn, S = 2, 60
    sd = {n: (p.clone().requires_grad_(True) if p.is_floating_point() else p) for n, p in O.synth_state_dict(dn=dn, seq_len=S).items()}
    x = torch.stack([O.synth_xT(0, dn * S)])
    cond = torch.stack([O.synth_cond(0, S)])
    t = torch.tensor([37])
    keep = torch.ones(1, dtype=torch.bool)
    a = O.decoder_forward(sd, x, cond, t, keep_mask=keep)
    with O.operand_rounding(fwd=True, bwd=True, out=True):
        b = O.decoder_forward(sd, x, cond, t, keep_mask=keep)
        (b ** 2).mean().backward()
    c = O.decoder_forward(sd, x, cond, t, keep_mask=keep)
    d = float((a - b).abs().max())
    assert torch.equal(a, c) and 1e-5 < d < 1e-1, d
    g = sd["seqTransDecoder.stack.3.linear1.weight"].grad
    assert g is not None and bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
