"""Data-format glue on either side of the hot path: the min-max ``Normalizer`` the samplers'
output is un-normalised with (reference dataset/preprocess.py:28-43, dataset/scaler.py:20-83), the checkpoint dictionary
(TCDiff.py:266-274, load :70-74,113-120 incl. the ``module.`` prefix of multi-process runs) and the trajectory hand-off
``x_0`` (TCDiff.py:283-302, :526-556).  Host-side tensor plumbing, no kernels."""
from __future__ import annotations

import io
import pickle
from typing import Dict, Optional

import torch


def _handle_zeros_in_scale(scale, copy=True, constant_mask=None):
    if constant_mask is None:
        constant_mask = scale < 10 * torch.finfo(scale.dtype).eps
    if copy:
        scale = scale.clone()
    scale[constant_mask] = 1.0
    return scale


class MinMaxScaler:
    """dataset/scaler.py:20-83 (attribute names kept: pickled reference scalers load into this class)."""

    def __init__(self, feature_range=(0, 1), *, copy=True, clip=False):
        self.feature_range = feature_range
        self.copy = copy
        self.clip = clip

    def fit(self, X):
        feature_range = self.feature_range
        if feature_range[0] >= feature_range[1]:
            raise ValueError("Minimum of desired feature range must be smaller than maximum. Got %s." % str(feature_range))
        data_min = torch.min(X, axis=0)[0]
        data_max = torch.max(X, axis=0)[0]
        self.n_samples_seen_ = X.shape[0]
        data_range = data_max - data_min
        self.scale_ = (feature_range[1] - feature_range[0]) / _handle_zeros_in_scale(data_range, copy=True)
        self.min_ = feature_range[0] - data_min * self.scale_
        self.data_min_, self.data_max_, self.data_range_ = data_min, data_max, data_range
        return self

    def transform(self, X):          # in place, like the reference
        X *= self.scale_.to(X.device)
        X += self.min_.to(X.device)
        if self.clip:
            torch.clip(X, self.feature_range[0], self.feature_range[1], out=X)
        return X

    def inverse_transform(self, X):  # in place; a narrower X uses the LAST columns of the fitted ones (scaler.py:79-82)
        X -= self.min_[-X.shape[1]:].to(X.device)
        X /= self.scale_[-X.shape[1]:].to(X.device)
        return X


class Normalizer:
    """dataset/preprocess.py:28-43."""

    def __init__(self, data):
        flat = data.reshape(-1, data.shape[-1])
        self.scaler = MinMaxScaler((-1, 1), clip=True)
        self.scaler.fit(flat)

    def normalize(self, x):
        batch, seq, ch = x.shape
        x = x.reshape(-1, ch)
        return self.scaler.transform(x).reshape((batch, seq, ch))

    def unnormalize(self, x):
        batch, seq, ch = x.shape
        x = x.reshape(-1, ch)
        x = torch.clip(x, -1, 1)  # clip to force compatibility
        return self.scaler.inverse_transform(x).reshape((batch, seq, ch))


# ---- checkpoints -------------------------------------------------------------------------------------------------------
class _RefUnpickler(pickle.Unpickler):
    """Reference checkpoints pickle `dataset.preprocess.Normalizer` / `dataset.scaler.MinMaxScaler` instances
    (TCDiff.py:270): map them to the classes above so a checkpoint loads without the reference on the path."""
    _MAP = {("dataset.preprocess", "Normalizer"): Normalizer, ("dataset.scaler", "MinMaxScaler"): MinMaxScaler}

    def find_class(self, module, name):
        if (module, name) in self._MAP:
            return self._MAP[(module, name)]
        return super().find_class(module, name)


class _RefPickle:
    __name__ = "tcdiff_amd.io._RefPickle"
    Unpickler = _RefUnpickler
    load = staticmethod(lambda f, **kw: _RefUnpickler(f, **kw).load())
    loads = staticmethod(lambda b, **kw: _RefUnpickler(io.BytesIO(b), **kw).load())
    dump, dumps, Pickler = pickle.dump, pickle.dumps, pickle.Pickler


def wrap(x: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {f"module.{key}": value for key, value in x.items()}          # TCDiff.py:31-33


def maybe_wrap(x, num):
    return x if num == 1 else wrap(x)                                    # TCDiff.py:35-36


def save_checkpoint(path: str, diffusion, model, optim, normalizer) -> None:
    """The four-key dictionary of TCDiff.py:266-273."""
    torch.save({"ema_state_dict": diffusion.master_model.state_dict(), "model_state_dict": model.state_dict(),
                "optimizer_state_dict": optim.state_dict(), "normalizer": normalizer}, path)


def load_checkpoint(path: str, model: Optional[torch.nn.Module] = None, EMA: bool = True, map_location="cpu"):
    """torch.load of a reference (or our) checkpoint; with `model`, loads the EMA (or raw) weights with strict=False as
    TCDiff.py:113-120 does -- keys saved with the `module.` prefix of a multi-process run are accepted as well."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_RefPickle)
    if model is not None:
        sd = ckpt["ema_state_dict" if EMA else "model_state_dict"]
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        model.load_state_dict(sd, strict=False)
    return ckpt


# ---- trajectory hand-off -------------------------------------------------------------------------------------------------
def x0_from_motion(x: torch.Tensor) -> torch.Tensor:
    """x (bs, dn, seq, 151) in the dataset layout -> x_0 (bs, seq * dn, 3): the xy trajectory (channels 4, 5) in a
    zero-padded xyz triple, token order frame-major (TCDiff.py:283-302)."""
    xy = x[:, :, :, [4, 5]]
    bs, dn, seq, _ = xy.shape
    traj = torch.zeros(bs, dn, seq, 3).to(xy)
    traj[:, :, :, [0, 1]] = xy[:, :, :, [0, 1]]
    return traj.permute(0, 2, 1, 3).reshape(bs, seq * dn, 3)


def x0_from_trajectory(traj_xy: torch.Tensor) -> torch.Tensor:
    """(bs, dn, seq, 2) predicted xy trajectories (the Dance-Beat Navigator's output, TCDiff.py:526-556) -> x_0."""
    bs, dn, seq, _ = traj_xy.shape
    traj = torch.zeros(bs, dn, seq, 3).to(traj_xy)
    traj[..., :2] = traj_xy
    return traj.permute(0, 2, 1, 3).reshape(bs, seq * dn, 3)
