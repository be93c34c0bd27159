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
