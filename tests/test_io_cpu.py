"""tcdiff_amd/io.py against fixtures produced by the reference's own classes (tests/golden/make_golden_io.py)."""
import os

import numpy as np
import torch
import torch.nn.functional as F

from tcdiff_amd import io as IO
from tcdiff_amd.model import DanceDecoder


def test_normalizer_matches_reference(golden_dir):
    ref = np.load(os.path.join(golden_dir, "normalizer.npz"))
    n = IO.Normalizer(torch.from_numpy(ref["data"]).clone())
    assert np.array_equal(n.normalize(torch.from_numpy(ref["x"]).clone()).numpy(), ref["xn"])
    assert np.array_equal(n.unnormalize(torch.from_numpy(ref["y"]).clone()).numpy(), ref["yu"])
    assert np.array_equal(n.unnormalize(torch.from_numpy(ref["y147"]).clone()).numpy(), ref["yu147"])
    assert float(np.abs(ref["xn"]).max()) <= 1.0            # clip=True


def test_reference_checkpoint_loads_without_the_reference(golden_dir):
    model = DanceDecoder(nfeats=151, seq_len=150, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                         cond_feature_dim=438, activation=F.gelu, required_dancer_num=3)
    ckpt = IO.load_checkpoint(os.path.join(golden_dir, "ref_ckpt_small.pt"), model, EMA=True)
    assert set(ckpt) == {"ema_state_dict", "model_state_dict", "optimizer_state_dict", "normalizer"}
    assert isinstance(ckpt["normalizer"], IO.Normalizer) and isinstance(ckpt["normalizer"].scaler, IO.MinMaxScaler)
    assert torch.equal(model.final_layer.bias.detach(), torch.arange(151.0))          # module.-prefixed keys accepted
    assert float(model.null_cond_hidden.detach().mean()) == 0.5
    ref = np.load(os.path.join(golden_dir, "normalizer.npz"))
    got = ckpt["normalizer"].unnormalize(torch.from_numpy(ref["y"]).clone()).numpy()
    assert np.array_equal(got, ref["yu"])


def test_checkpoint_round_trip_and_x0(tmp_path):
    from tcdiff_amd import Adan
    from tcdiff_amd.diffusion import GaussianDiffusion
    model = DanceDecoder(nfeats=151, seq_len=60, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                         cond_feature_dim=438, activation=F.gelu, required_dancer_num=2)
    diff = GaussianDiffusion(model, 60, 151, None, schedule="cosine", n_timestep=100, predict_epsilon=False,
                             loss_type="l2", cond_drop_prob=0.25, guidance_weight=2, seq_len=60)
    opt = Adan(model.parameters(), lr=5e-5, weight_decay=0.02)
    norm = IO.Normalizer(torch.randn(3, 10, 151))
    p = str(tmp_path / "train-1.pt")
    IO.save_checkpoint(p, diff, model, opt, norm)
    m2 = DanceDecoder(nfeats=151, seq_len=60, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                      cond_feature_dim=438, activation=F.gelu, required_dancer_num=2)
    ck = IO.load_checkpoint(p, m2, EMA=False)
    assert all(torch.equal(a, b) for a, b in zip(model.state_dict().values(), m2.state_dict().values()))
    assert isinstance(ck["normalizer"], IO.Normalizer) and torch.equal(ck["normalizer"].scaler.scale_, norm.scaler.scale_)
    # the file names the REFERENCE's classes, so the reference's plain torch.load reads it like one of its own
    import zipfile
    raw = zipfile.ZipFile(p).read([n for n in zipfile.ZipFile(p).namelist() if n.endswith("data.pkl")][0])
    assert b"dataset.preprocess\nNormalizer" in raw and b"dataset.scaler\nMinMaxScaler" in raw and b"tcdiff_amd" not in raw
    assert IO.maybe_wrap({"a": 1}, 2) == {"module.a": 1} and IO.maybe_wrap({"a": 1}, 1) == {"a": 1}
    x = torch.randn(2, 3, 150, 151)
    x0 = IO.x0_from_motion(x)
    assert x0.shape == (2, 450, 3) and float(x0[..., 2].abs().max()) == 0
    assert torch.equal(x0.reshape(2, 150, 3, 3)[:, :, 1, :2], x[:, 1, :, 4:6])
    assert torch.equal(IO.x0_from_trajectory(x[..., 4:6]), x0)


def test_checkpoint_loader_refuses_globals_outside_the_allow_list(tmp_path):
    import pickle
    import pytest

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("true",))
    p = str(tmp_path / "evil.pt")
    torch.save({"normalizer": Evil()}, p)
    with pytest.raises(pickle.UnpicklingError):
        IO.load_checkpoint(p)


def test_checkpoint_loader_refuses_dotted_names_inside_an_allowed_module(tmp_path):
    """protocol 4's STACK_GLOBAL resolves a dotted name attribute by attribute, so ("torch", "os.getcwd") walked
    out of the allow-listed module.  Hand-written pickles: they must raise, not call."""
    import pickle
    import pytest
    for module, name in (("torch", "os.getcwd"), ("torch", "os.system"), ("builtins", "getattr"), ("torch.serialization", "os.getcwd"),
                         ("numpy", "os.getcwd"), ("collections", "_sys.exit")):
        payload = (pickle.PROTO + b"\x04" + pickle.SHORT_BINUNICODE + bytes([len(module)]) + module.encode() +
                   pickle.SHORT_BINUNICODE + bytes([len(name)]) + name.encode() + pickle.STACK_GLOBAL + pickle.EMPTY_TUPLE +
                   pickle.REDUCE + pickle.STOP)
        with pytest.raises(pickle.UnpicklingError):
            IO._RefPickle.loads(payload)
    # ... and the names a real checkpoint needs still resolve
    assert IO._RefPickle.loads(pickle.dumps(__import__("collections").OrderedDict(a=1))) == {"a": 1}


def _kalman_loop(xy, dt=1.0, q=1e-2, r=1e-1):
    """Per-trajectory transcription of filterpy 1.4.5's predict / update as the reference drives them
    (TrajDecoder/utils/utils_model.py:25-72): the independent check of io.kalman_smooth_batch's batched form."""
    import numpy as np
    out = np.zeros_like(xy)
    for b in range(xy.shape[0]):
        for d in range(xy.shape[1]):
            F = np.array([[1, 0, dt, 0], [0, 1, 0, dt], [0, 0, 1, 0], [0, 0, 0, 1.0]])
            H = np.array([[1, 0, 0, 0], [0, 1, 0, 0.0]])
            P, R, Q = np.eye(4) * 10.0, np.eye(2) * r ** 2, np.eye(4) * q
            x = np.array([[xy[b, d, 0, 0]], [xy[b, d, 0, 1]], [0.0], [0.0]])
            for t in range(xy.shape[2]):
                x = F.dot(x)
                P = F.dot(P).dot(F.T) + Q
                z = xy[b, d, t].reshape(2, 1)
                y = z - H.dot(x)
                PHT = P.dot(H.T)
                S = H.dot(PHT) + R
                K = PHT.dot(np.linalg.inv(S))
                x = x + K.dot(y)
                I_KH = np.eye(4) - K.dot(H)
                P = I_KH.dot(P).dot(I_KH.T) + K.dot(R).dot(K.T)
                out[b, d, t] = x[:2, 0]
    return out


def test_kalman_smooth_batch_against_the_per_trajectory_loop():
    import numpy as np
    rng = np.random.default_rng(5)
    for dtype in (np.float32, np.float64):
        steps = rng.normal(0, 0.05, (3, 4, 160, 2)).cumsum(axis=2)
        xy = (steps + rng.normal(0, 0.1, steps.shape)).astype(dtype)
        got = IO.kalman_smooth_batch(xy)
        assert got.shape == xy.shape and got.dtype == xy.dtype
        want = _kalman_loop(xy)
        assert np.abs(got.astype(np.float64) - want.astype(np.float64)).max() <= (1e-12 if dtype == np.float64 else 1e-6)
    # other arguments, one frame, empty
    xy = rng.normal(size=(1, 2, 30, 2))
    assert np.allclose(IO.kalman_smooth_batch(xy, dt=0.5, process_noise_std=0.1, measurement_noise_std=0.3),
                       _kalman_loop(xy, 0.5, 0.1, 0.3), atol=1e-12)
    assert np.allclose(IO.kalman_smooth_batch(xy[:, :, :1]), xy[:, :, :1], atol=1e-3)   # first frame: gain ~ 1 (P0 = 10 >> R)
    assert IO.kalman_smooth_batch(np.zeros((0, 3, 5, 2))).shape == (0, 3, 5, 2)
    import pytest
    with pytest.raises(ValueError):
        IO.kalman_smooth_batch(np.zeros((2, 5, 3)))


def test_kalman_properties():
    """What a constant-velocity filter must do whatever the library: a stationary dancer stays put, a constant-velocity
    one is tracked with vanishing lag, measurement noise shrinks, and the filter is linear in the trajectory."""
    import numpy as np
    rng = np.random.default_rng(11)
    still = np.tile(np.array([0.3, -0.7]), (2, 3, 150, 1))
    assert np.abs(IO.kalman_smooth_batch(still) - still).max() < 1e-9
    t = np.arange(150.0)[None, None, :, None]
    line = np.concatenate([0.01 * t + 0.2, -0.02 * t + 0.1], -1)
    sm = IO.kalman_smooth_batch(line)
    assert np.abs(sm[0, 0, 40:] - line[0, 0, 40:]).max() < 2e-3
    noise = rng.normal(0, 0.1, (4, 3, 150, 2))
    a = rng.normal(size=(4, 3, 150, 2)).cumsum(2) * 0.02
    assert np.allclose(IO.kalman_smooth_batch(a + 2.0 * noise), IO.kalman_smooth_batch(a) + 2.0 * IO.kalman_smooth_batch(noise), atol=1e-9)
    ns = IO.kalman_smooth_batch(line + noise[:1, :1])
    # Q = 1e-2 I against R = 1e-2 I is a light filter (the reference passes the process STD as a variance): ~14 % less noise
    assert np.sqrt(((ns[0, 0, 20:] - line[0, 0, 20:]) ** 2).mean()) < 0.95 * np.sqrt((noise[0, 0, 20:] ** 2).mean())


def test_navigator_hand_off_end_to_end():
    """TCDiff.py:543-556: predicted xy (bs, dn, seq, 2) -> smoothing -> zero-z padding -> x_0 (bs, seq dn, 3) frame-major."""
    import numpy as np
    torch.manual_seed(0)
    traj = torch.randn(2, 3, 150, 2).cumsum(2) * 0.02
    x0 = IO.x0_from_navigator(traj)
    assert x0.shape == (2, 450, 3) and x0.dtype == traj.dtype and float(x0[..., 2].abs().max()) == 0
    # the reference's own lines, written out
    sm = torch.from_numpy(IO.kalman_smooth_batch(traj.cpu().detach().numpy())).to(dtype=traj.dtype, device=traj.device)
    pad = torch.zeros(2, 3, 150, 3).to(sm)
    pad[:, :, :, [0, 1]] = sm[:, :, :, [0, 1]]
    assert torch.equal(x0, pad.permute(0, 2, 1, 3).reshape(2, 450, 3))
    assert torch.equal(IO.x0_from_navigator(traj, smooth=False), IO.x0_from_trajectory(traj))
    assert not np.allclose(x0.numpy(), IO.x0_from_trajectory(traj).numpy())
