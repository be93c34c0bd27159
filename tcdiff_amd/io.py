"""Data-format glue on either side of the hot path: the min-max ``Normalizer`` the samplers'
output is un-normalised with (reference dataset/preprocess.py:28-43, dataset/scaler.py:20-83), the checkpoint dictionary
(TCDiff.py:266-274, load :70-74,113-120 incl. the ``module.`` prefix of multi-process runs) and the trajectory hand-off
``x_0`` (TCDiff.py:283-302, :526-556).  Host-side tensor plumbing, no kernels."""
from __future__ import annotations

import io
import pickle
from typing import Dict, Optional

import torch


class MinMaxScaler:
    """Column-wise affine map onto ``feature_range`` fitted from the column minima and maxima -- the arithmetic of the
    reference's dataset/scaler.py:20-83 (`x * scale_ + min_`, inverse `(x - min_) / scale_`, both IN PLACE), with its
    attribute names, because reference checkpoints pickle instances of it (TCDiff.py:270) and they load into this class."""

    def __init__(self, feature_range=(0, 1), *, copy=True, clip=False):
        self.feature_range, self.copy, self.clip = feature_range, copy, clip

    def fit(self, X):
        lo, hi = self.feature_range
        if not lo < hi:
            raise ValueError(f"feature_range must be increasing, got {self.feature_range}")
        cmin, cmax = X.amin(dim=0), X.amax(dim=0)
        span = cmax - cmin
        # a (near-)constant column gets a unit divisor instead of a division by ~0
        divisor = torch.where(span < 10 * torch.finfo(span.dtype).eps, torch.ones_like(span), span)
        self.scale_ = (hi - lo) / divisor
        self.min_ = lo - cmin * self.scale_
        self.data_min_, self.data_max_, self.data_range_, self.n_samples_seen_ = cmin, cmax, span, X.shape[0]
        return self

    def transform(self, X):
        X.mul_(self.scale_.to(X.device)).add_(self.min_.to(X.device))
        return X.clamp_(*self.feature_range) if self.clip else X

    def inverse_transform(self, X):
        # a narrower X is taken to be the LAST columns of the fitted ones (the 147-wide motion without contacts)
        n = X.shape[1]
        return X.sub_(self.min_[-n:].to(X.device)).div_(self.scale_[-n:].to(X.device))


class Normalizer:
    """[-1, 1] min-max normaliser over the last axis (reference dataset/preprocess.py:28-43); `.scaler` as pickled there."""

    def __init__(self, data):
        self.scaler = MinMaxScaler((-1, 1), clip=True).fit(data.reshape(-1, data.shape[-1]))

    def normalize(self, x):
        return self.scaler.transform(x.reshape(-1, x.shape[-1])).reshape(x.shape)

    def unnormalize(self, x):
        # values are forced into the fitted range first; the clamp makes the copy the inverse map then overwrites
        return self.scaler.inverse_transform(x.reshape(-1, x.shape[-1]).clamp(-1, 1)).reshape(x.shape)


# ---- checkpoints -------------------------------------------------------------------------------------------------------
_REF_NAMES = {("dataset.preprocess", "Normalizer"): Normalizer, ("dataset.scaler", "MinMaxScaler"): MinMaxScaler}
_SAFE_BUILTINS = {"set", "frozenset", "list", "dict", "tuple", "int", "float", "bool", "complex", "str", "bytes", "slice",
                  "range", "bytearray", "getattr"}


class _RefUnpickler(pickle.Unpickler):
    """Reference checkpoints pickle `dataset.preprocess.Normalizer` / `dataset.scaler.MinMaxScaler` instances
    (TCDiff.py:270): they are mapped to the classes above, so a checkpoint loads without the reference on the path.
    Everything else is held to an allow-list (torch's own rebuild helpers, collections, numpy array reconstruction, a few
    builtins): a checkpoint is data, and an arbitrary global in it is refused rather than imported."""

    def find_class(self, module, name):
        if (module, name) in _REF_NAMES:
            return _REF_NAMES[(module, name)]
        ok = (module == "torch" or module.startswith("torch.") or module == "collections"
              or module in ("numpy", "numpy.core.multiarray", "numpy._core.multiarray", "numpy.core.numeric", "numpy._core.numeric")
              or (module == "builtins" and name in _SAFE_BUILTINS)
              or (module == "tcdiff_amd.io" and name in ("Normalizer", "MinMaxScaler")))
        if not ok:
            raise pickle.UnpicklingError(f"checkpoint refers to {module}.{name}, which is not on the allow-list")
        return super().find_class(module, name)


class _RefPickler(pickle._Pickler):
    """Writes the two classes above under the REFERENCE's module paths, so that a checkpoint saved here is loaded by the
    reference's plain `torch.load` (TCDiff.py:70-74) exactly like one of its own."""
    _OUT = {cls: mod_name for mod_name, cls in _REF_NAMES.items()}

    def save_global(self, obj, name=None):
        if obj in self._OUT:
            module, qual = self._OUT[obj]
            self.write(pickle.GLOBAL + f"{module}\n{qual}\n".encode("ascii"))
            self.memoize(obj)
            return
        super().save_global(obj, name)

    dispatch = dict(pickle._Pickler.dispatch)
    dispatch[type] = save_global


class _RefPickle:
    __name__ = "tcdiff_amd.io._RefPickle"
    Unpickler, Pickler = _RefUnpickler, _RefPickler
    load = staticmethod(lambda f, **kw: _RefUnpickler(f, **kw).load())
    loads = staticmethod(lambda b, **kw: _RefUnpickler(io.BytesIO(b), **kw).load())

    @staticmethod
    def dump(obj, f, protocol=None, **kw):
        _RefPickler(f, protocol).dump(obj)

    @staticmethod
    def dumps(obj, protocol=None, **kw):
        buf = io.BytesIO()
        _RefPickler(buf, protocol).dump(obj)
        return buf.getvalue()


def wrap(x: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {f"module.{key}": value for key, value in x.items()}          # TCDiff.py:31-33


def maybe_wrap(x, num):
    return x if num == 1 else wrap(x)                                    # TCDiff.py:35-36


def save_checkpoint(path: str, diffusion, model, optim, normalizer) -> None:
    """The four-key dictionary of TCDiff.py:266-273; the normalizer is written under the reference's class paths, so the
    file is loadable by the reference as well as by `load_checkpoint`."""
    torch.save({"ema_state_dict": diffusion.master_model.state_dict(), "model_state_dict": model.state_dict(),
                "optimizer_state_dict": optim.state_dict(), "normalizer": normalizer}, path, pickle_module=_RefPickle)


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
