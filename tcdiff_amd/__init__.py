"""tcdiff_amd: MI355X-native (gfx950) implementation of TCDiff's denoising hot path.

    from tcdiff_amd import DanceDecoder, GaussianDiffusion     # drop-ins for model.model / model.diffusion
"""
from .model import DanceDecoder  # noqa: F401
from .diffusion import GaussianDiffusion, EMA  # noqa: F401
from .adan import Adan  # noqa: F401
from .fk import SMPLSkeleton, ax_from_6v  # noqa: F401

__all__ = ["DanceDecoder", "GaussianDiffusion", "EMA", "Adan", "SMPLSkeleton", "ax_from_6v"]
