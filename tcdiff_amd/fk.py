"""SMPL forward kinematics and the 6-D rotation conversion on MI355X -- counterparts of the reference's
``vis.SMPLSkeleton`` (vis.py:330-406) and ``dataset.quaternion.ax_from_6v`` (dataset/quaternion.py:28-32), which the
training loss calls at model/diffusion.py:693-708.  One fused kernel each (csrc/train.hip); the reference goes through
pytorch3d (absent here: arithmetic restated from its published definitions, cross-checked against scipy in
tests/test_train_cpu.py -- "parity unpinned").
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import kernels as K

# vis.py:48-73 / :76-101
SMPL_PARENTS = [-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21]
SMPL_OFFSETS = [
    [0.0, 0.0, 0.0], [0.05858135, -0.08228004, -0.01766408], [-0.06030973, -0.09051332, -0.01354254],
    [0.00443945, 0.12440352, -0.03838522], [0.04345142, -0.38646945, 0.008037],
    [-0.04325663, -0.38368791, -0.00484304], [0.00448844, 0.1379564, 0.02682033],
    [-0.01479032, -0.42687458, -0.037428], [0.01905555, -0.4200455, -0.03456167],
    [-0.00226458, 0.05603239, 0.00285505], [0.04105436, -0.06028581, 0.12204243],
    [-0.03483987, -0.06210566, 0.13032329], [-0.0133902, 0.21163553, -0.03346758],
    [0.07170245, 0.11399969, -0.01889817], [-0.08295366, 0.11247234, -0.02370739],
    [0.01011321, 0.08893734, 0.05040987], [0.12292141, 0.04520509, -0.019046],
    [-0.11322832, 0.04685326, -0.00847207], [0.2553319, -0.01564902, -0.02294649],
    [-0.26012748, -0.01436928, -0.03126873], [0.26570925, 0.01269811, -0.00737473],
    [-0.26910836, 0.00679372, -0.00602676], [0.08669055, -0.01063603, -0.01559429],
    [-0.0887537, -0.00865157, -0.01010708]]


def _cuda_f32(t: torch.Tensor, what: str) -> torch.Tensor:
    if not t.is_cuda:
        raise L.TcdiffError(f"{what} runs on MI355X only (no CPU fallback; the CPU oracle lives in oracle/)")
    return t.float().contiguous()


def ax_from_6v(q: torch.Tensor) -> torch.Tensor:
    """(..., 6) continuous 6-D rotations -> (..., 3) axis-angle (dataset/quaternion.py:28-32)."""
    assert q.shape[-1] == 6
    q = _cuda_f32(q, "ax_from_6v")
    out = torch.empty(q.shape[:-1] + (3,), device=q.device, dtype=torch.float32)
    K.ax_from_6v(q, q.numel() // 6, 1, 6, out)
    return out


class SMPLSkeleton:
    """Same constructor and ``forward(rotations (N, L, J, 3) axis-angle, root_positions (N, L, 3)) -> (N, L, J, 3)``
    as vis.SMPLSkeleton (vis.py:330-406)."""

    def __init__(self, device=None, offsets=None, parents=None):
        self._parents = list(SMPL_PARENTS if parents is None else parents)
        self._offsets_list = [list(map(float, o)) for o in (SMPL_OFFSETS if offsets is None else offsets)]
        assert len(self._offsets_list) == len(self._parents) == 24
        self._offsets = torch.tensor(self._offsets_list, device=device)

    def forward(self, rotations: torch.Tensor, root_positions: torch.Tensor) -> torch.Tensor:
        assert len(rotations.shape) == 4, "Rotations should be a 4D tensor."
        assert len(root_positions.shape) == 3, "Root positions should be a 3D tensor."
        rot, root = _cuda_f32(rotations, "SMPLSkeleton.forward"), _cuda_f32(root_positions, "SMPLSkeleton.forward")
        N, Lq = rot.shape[:2]
        out = torch.empty(N, Lq, 24, 3, device=rot.device, dtype=torch.float32)
        K.smpl_fk(rot, root, N * Lq, self._parents, self._offsets_list, out)
        return out
