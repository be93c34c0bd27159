"""Import the REAL reference (Da1yuqin/TCDiff) from /root/reference -- build container only.

Test infrastructure: used by tests/golden/make_golden*.py to
pin the oracle and to generate golden vectors.  /root/reference does not exist on the GPU box, so nothing
in ``-m gpu`` tests, smoke() or bench.py may call this.

The reference's model/diffusion.py imports rendering/FK helpers (p_tqdm, pytorch3d, librosa, soundfile)
that are absent here and are never touched by the sampling path; empty placeholder modules satisfy the
import statements only (any call into them raises).
"""
import os
import sys
import types

REF = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "model"))


def _placeholder(name, names=()):
    def _unavailable(*a, **k):
        raise NotImplementedError(f"{name} is a placeholder; not available in this container")
    m = types.ModuleType(name)
    for n in names:
        setattr(m, n, _unavailable)
    sys.modules.setdefault(name, m)


def load():
    """Returns (DanceDecoder, GaussianDiffusion) classes of the real reference."""
    if not available():
        raise RuntimeError("/root/reference is not present on this box")
    os.environ.setdefault("TQDM_DISABLE", "1")
    _placeholder("p_tqdm", ["p_map"])
    _placeholder("pytorch3d")
    _placeholder("librosa")
    _placeholder("soundfile")
    _placeholder("pytorch3d.transforms", [
        "axis_angle_to_quaternion", "quaternion_to_axis_angle", "quaternion_apply", "quaternion_multiply",
        "axis_angle_to_matrix", "matrix_to_axis_angle", "matrix_to_quaternion", "matrix_to_rotation_6d",
        "quaternion_to_matrix", "rotation_6d_to_matrix", "RotateAxisAngle"])
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from model.model import DanceDecoder  # noqa: E402
    from model.diffusion import GaussianDiffusion  # noqa: E402
    return DanceDecoder, GaussianDiffusion


def build_reference(sd, dn=3, seq_len=150, n_timestep=1000, latent=512, ff=1024, n_layers=8, n_head=8,
                    cond_dim=438, nfeats=151, guidance_weight=2, activation=None, use_rotary=True):
    """Reference model + diffusion with the production ctor arguments (TCDiff.py:76-102), eval mode."""
    import torch.nn.functional as F
    DanceDecoder, GaussianDiffusion = load()
    model = DanceDecoder(nfeats=nfeats, seq_len=seq_len, latent_dim=latent, ff_size=ff, num_layers=n_layers,
                         num_heads=n_head, dropout=0.1, cond_feature_dim=cond_dim,
                         activation=F.gelu if activation is None else activation, required_dancer_num=dn, use_rotary=use_rotary)
    missing, unexpected = model.load_state_dict(sd, strict=True), None
    model.eval()
    diff = GaussianDiffusion(model, seq_len, nfeats, None, schedule="cosine", n_timestep=n_timestep,
                             predict_epsilon=False, loss_type="l2", use_p2=False, cond_drop_prob=0.25,
                             guidance_weight=guidance_weight, seq_len=seq_len)
    diff.eval()
    return model, diff


class patched_randn:
    """Context manager: torch.randn_like / torch.randn draw from a supplied per-call function
    (noise injection recipe)."""

    def __init__(self, like_fn=None, randn_fn=None):
        self.like_fn, self.randn_fn = like_fn, randn_fn

    def __enter__(self):
        import torch
        self._rl, self._r = torch.randn_like, torch.randn
        if self.like_fn is not None:
            torch.randn_like = lambda t, *a, **k: self.like_fn(t)
        if self.randn_fn is not None:
            # only the reference's generator-less draws are replaced; seeded draws (our own input
            # recipes) go to the real torch.randn
            torch.randn = lambda *a, **k: self._r(*a, **k) if "generator" in k else self.randn_fn(*a, **k)
        return self

    def __exit__(self, *exc):
        import torch
        torch.randn_like, torch.randn = self._rl, self._r
