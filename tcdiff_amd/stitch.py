"""Long-sequence generation across ranks and the window stitch.

* ``halo_exchange``: ``long_ddim_sample`` / ``long_inpaint_loop`` generate half-overlapping windows as a batch and, after
  every step, copy each window's second half into the next window's first half (reference model/diffusion.py:502-506,
  599-601).  When the windows are sharded over GPUs in contiguous blocks (tcdiff_amd/dist.py) the copy inside a rank is
  the captured ``tcdiff_window_couple_step`` launch and ONE boundary crosses each rank pair: rank r sends its last
  window's second half to rank r + 1 (point-to-point over RCCL/xGMI, 135.9 KB at 3 dancers x 150 frames).
* ``stitch_windows``: the render-time merge of the windows into one sequence (reference model/diffusion.py:841-897):
  root positions cross-faded linearly, joint rotations slerped (dataset/quaternion.py:35-71).  Plain tensor arithmetic on
  the result tensors (host plumbing, no kernel): the axis-angle <-> quaternion conversions restate pytorch3d's
  definitions ("parity unpinned", cross-checked against scipy in tests/test_stitch_cpu.py).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


# ----------------------------------------------------------------------------------------------------------------------
def halo_exchange(x: torch.Tensor, seq_len: int, row_elems: int) -> None:
    """x: this rank's windows (b_local, L, nfeat), contiguous, L * nfeat == seq_len * row_elems.  In place:
    first half of this rank's FIRST window <- second half of the previous rank's LAST window."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    rank, world = dist.get_rank(), dist.get_world_size()
    if x.shape[0] == 0:
        raise ValueError("halo_exchange: a rank without windows breaks the neighbour chain (shard n_windows >= world)")
    half = seq_len // 2
    xv = x.view(x.shape[0], seq_len, row_elems)
    ops, recv = [], None
    if rank + 1 < world:
        ops.append(dist.P2POp(dist.isend, xv[-1, half:].contiguous(), rank + 1))
    if rank > 0:
        recv = torch.empty(half, row_elems, device=x.device, dtype=x.dtype)
        ops.append(dist.P2POp(dist.irecv, recv, rank - 1))
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    if recv is not None:
        xv[0, :half] = recv


# ----------------------------------------------------------------------------------------------------------------------
def _axis_angle_to_quaternion(aa: torch.Tensor) -> torch.Tensor:
    ang = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = ang * 0.5
    small = ang.abs() < 1e-6
    k = torch.where(small, 0.5 - (ang * ang) / 48, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    return torch.cat([torch.cos(half), aa * k], dim=-1)


def _quaternion_to_axis_angle(q: torch.Tensor) -> torch.Tensor:
    norms = torch.norm(q[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    ang = 2 * half
    small = ang.abs() < 1e-6
    k = torch.where(small, 0.5 - (ang * ang) / 48, torch.sin(half) / torch.where(small, torch.ones_like(ang), ang))
    return q[..., 1:] / k


def quat_slerp(x: torch.Tensor, y: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
    """dataset/quaternion.py:35-71 (without its in-place writes into the arguments)."""
    ln = torch.sum(x * y, dim=-1)
    neg = ln < 0.0
    ln = torch.where(neg, -ln, ln)
    y = torch.where(neg[..., None], -y, y)
    a = torch.zeros_like(x[..., 0]) + a
    linear = (1.0 - ln) < 0.01
    om = torch.arccos(torch.where(linear, torch.zeros_like(ln), ln))
    so = torch.where(linear, torch.ones_like(om), torch.sin(om))
    a0 = torch.where(linear, 1.0 - a, torch.sin((1.0 - a) * om) / so)
    a1 = torch.where(linear, a, torch.sin(a * om) / so)
    return a0[..., None] * x + a1[..., None] * y


def stitch_windows(pos: torch.Tensor, q: torch.Tensor):
    """pos (b, s, dn, 3) root positions and q (b, s, dn, J, 3) axis-angle rotations of b half-overlapping windows ->
    (s + (b - 1) s / 2, dn, 3) and (.., dn, J, 3) (reference model/diffusion.py:841-897)."""
    b, s, dn = pos.shape[:3]
    assert s % 2 == 0
    half = s // 2
    total = s + half * (b - 1)
    fade_out = torch.ones(1, s, 1, 1, dtype=pos.dtype, device=pos.device)
    fade_in = torch.ones_like(fade_out)
    fade_out[:, half:] = torch.linspace(1, 0, half, dtype=pos.dtype, device=pos.device)[None, :, None, None]
    fade_in[:, :half] = torch.linspace(0, 1, half, dtype=pos.dtype, device=pos.device)[None, :, None, None]
    p = pos.clone()
    p[:-1] *= fade_out
    p[1:] *= fade_in
    full_pos = torch.zeros(total, dn, 3, dtype=pos.dtype, device=pos.device)
    for i in range(b):
        full_pos[i * half:i * half + s] += p[i]
    full_q = torch.zeros((total,) + tuple(q.shape[2:]), dtype=q.dtype, device=q.device)
    full_q[:half] = q[0, :half]
    if b > 1:
        w = torch.linspace(0, 1, half, dtype=q.dtype, device=q.device)[None, :, None, None]
        merged = _quaternion_to_axis_angle(quat_slerp(_axis_angle_to_quaternion(q[:-1, half:]),
                                                      _axis_angle_to_quaternion(q[1:, :half]), w))
        for i in range(b - 1):
            full_q[half + i * half: half + (i + 1) * half] = merged[i]
    full_q[total - half:] = q[-1, half:]
    return full_pos, full_q
