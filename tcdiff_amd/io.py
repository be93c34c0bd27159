"""Data-format glue on either side of the hot path: the min-max ``Normalizer`` the samplers'
output is un-normalised with (reference dataset/preprocess.py:28-43, dataset/scaler.py:20-83), the checkpoint dictionary
(TCDiff.py:266-274, load :70-74,113-120 incl. the ``module.`` prefix of multi-process runs) and the trajectory hand-off
``x_0`` (TCDiff.py:283-302, :526-556) including the Kalman smoothing of the predicted trajectories (TCDiff.py:546,
TrajDecoder/utils/utils_model.py:10-74).  Host-side tensor plumbing, no kernels."""
from __future__ import annotations

import io
import pickle
from typing import Dict, Optional

import numpy as np
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
                  "range", "bytearray"}
# every global a checkpoint of TCDiff.py:266-273 (two state dicts, Adan's state, the normalizer) refers to, by exact name
_SAFE_GLOBALS = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch", "Size"), ("torch", "device"), ("torch", "dtype"), ("torch", "Tensor"), ("torch.nn.parameter", "Parameter"),
    ("torch.serialization", "_get_layout"), ("torch", "strided"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
    ("tcdiff_amd.io", "Normalizer"), ("tcdiff_amd.io", "MinMaxScaler"),
}
_SAFE_GLOBALS |= {("torch", f"{t}Storage") for t in ("Float", "Double", "Half", "BFloat16", "Long", "Int", "Short", "Char", "Byte",
                                                       "Bool", "Untyped")}
_SAFE_GLOBALS |= {("torch", t) for t in ("float32", "float64", "float16", "bfloat16", "int64", "int32", "int16", "int8", "uint8", "bool")}


class _RefUnpickler(pickle.Unpickler):
    """Reference checkpoints pickle `dataset.preprocess.Normalizer` / `dataset.scaler.MinMaxScaler` instances
    (TCDiff.py:270): they are mapped to the classes above, so a checkpoint loads without the reference on the path.
    Everything else is held to an EXACT (module, name) allow-list -- torch's tensor / storage rebuild helpers, collections,
    numpy array reconstruction, a few builtin types: a checkpoint is data, and any other global in it is refused rather than
    imported.  Dotted names are refused outright: protocol 4's STACK_GLOBAL resolves `name` attribute by attribute, so
    ("torch", "os.system") would otherwise walk out of an allowed module."""

    def find_class(self, module, name):
        if (module, name) in _REF_NAMES:
            return _REF_NAMES[(module, name)]
        ok = "." not in name and ((module, name) in _SAFE_GLOBALS or (module == "builtins" and name in _SAFE_BUILTINS))
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


def kalman_smooth_batch(xy_batch, dt=1.0, process_noise_std=1e-2, measurement_noise_std=1e-1):
    """Forward Kalman filter over every (clip, dancer) xy trajectory of `xy_batch` (batch, dancers, frames, 2) -> same shape
    and dtype: the smoothing step between the Dance-Beat Navigator's output and `x_0` (TCDiff.py:546; defined in
    TrajDecoder/utils/utils_model.py:10-74).  Constant-velocity model, state (x, y, vx, vy) started at the first sample with
    zero velocity, P0 = 10 I, Q = process_noise_std I (the reference passes the standard deviation itself, not its square,
    :49-53), R = measurement_noise_std^2 I; per frame one predict and one update, the filtered position is the output.

    The reference runs `filterpy.kalman.KalmanFilter` (filterpy==1.4.5, requirements.txt:69; NOT installed here and not
    under /root/reference: **parity unpinned**), whose published predict / update are written out below in its operation
    order: x = F x, P = F P F^T + Q; y = z - H x, S = H P H^T + R, K = P H^T S^-1, x += K y, P = (I - K H) P (I - K H)^T +
    K R K^T (Joseph form), all in float64.  The covariance recursion does not depend on the data, so P and K are advanced once
    per frame for ALL trajectories and only the state update is batched; a plain per-trajectory loop (tests/test_io_cpu.py)
    gives the same numbers."""
    xy = np.asarray(xy_batch)
    if xy.ndim != 4 or xy.shape[-1] != 2:
        raise ValueError(f"xy_batch must be (batch, dancers, frames, 2), got {xy.shape}")
    bs, dn, seq, _ = xy.shape
    out = np.zeros_like(xy)
    if seq == 0 or bs * dn == 0:
        return out
    F = np.array([[1, 0, dt, 0], [0, 1, 0, dt], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    H = np.array([[1, 0, 0, 0], [0, 1, 0, 0]], dtype=np.float64)
    R = np.eye(2) * measurement_noise_std ** 2
    Q = np.eye(4) * process_noise_std
    I4 = np.eye(4)
    P = np.eye(4) * 10.0
    z = xy.reshape(bs * dn, seq, 2)
    x = np.zeros((bs * dn, 4), dtype=np.float64)
    x[:, :2] = z[:, 0]
    for t in range(seq):
        x = x @ F.T                                   # predict
        P = F @ P @ F.T + Q
        y = z[:, t] - x @ H.T                         # update
        PHT = P @ H.T
        S = H @ PHT + R
        K = PHT @ np.linalg.inv(S)
        x = x + y @ K.T
        I_KH = I4 - K @ H
        P = I_KH @ P @ I_KH.T + K @ R @ K.T
        out.reshape(bs * dn, seq, 2)[:, t] = x[:, :2]
    return out


def x0_from_navigator(x_traj, smooth: bool = True) -> torch.Tensor:
    """The whole hand-off of TCDiff.py:543-556: predicted xy trajectories (bs, dn, seq, 2) as a tensor -> Kalman smoothing on
    the host in numpy -> back to the tensor's dtype / device -> zero z channel -> `x_0` (bs, seq * dn, 3) frame-major, the
    layout `render_sample(..., x_0=)` / `ddim_sample(x_0=)` take."""
    t = x_traj
    if smooth:
        sm = kalman_smooth_batch(t.detach().cpu().numpy())
        t = torch.from_numpy(sm).to(dtype=x_traj.dtype, device=x_traj.device)
    return x0_from_trajectory(t)
