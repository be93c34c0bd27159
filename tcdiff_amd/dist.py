"""Clip sharding over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" on CPU for tests).

Sampling is embarrassingly parallel over clips (every op of the denoiser and of p_sample is per clip:
reference model/model.py:548-624, model/diffusion.py:241-286), so the data path has NO collective inside the
step loop: clips are dealt to ranks in contiguous blocks, noise is keyed by the GLOBAL clip index
(tcdiff_sampler_update), and one all-gather collects the finished samples.  Weights are replicated: every rank
builds the same state_dict (by-name synthetic weights or the same checkpoint file)."""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def _force() -> bool:
    """TCDIFF_DIST_FORCE=1 (tests): a ONE-rank process group is treated like any other -- it is created, and gather / timing /
    barrier / gradient averaging go through their collectives instead of taking the world-size-1 shortcut.  RCCL accepts a
    one-rank communicator, so this is how every nccl-only branch of this module (ReduceOp.AVG, all_gather_into_tensor on
    device memory, stream-ordered wait()) runs on a one-GPU box (tests/test_rccl_world1_gpu.py)."""
    return os.environ.get("TCDIFF_DIST_FORCE", "0") == "1"


def collectives_on(group=None) -> bool:
    """whether the job's collectives are to be issued: a process group exists and has more than one rank (or _force())"""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or _force())


def init_from_env(backend: str | None = None) -> Tuple[int, int, int]:
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  A single process without the env vars is world_size 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or _force()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(n_clips: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of global clip indices owned by `rank` (blocks differ by at most one clip;
    contiguous so that long-mode window coupling only crosses one boundary per rank pair)."""
    base, rem = divmod(n_clips, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_samples(local: torch.Tensor, n_clips: int) -> torch.Tensor:
    """All-gather per-rank samples (b_local, L, F) into (n_clips, L, F) on every rank, in global clip order.
    Equal shards use one all_gather_into_tensor (a single RCCL collective); ragged shards pad to the largest."""
    if not collectives_on():
        return local
    world = dist.get_world_size()
    sizes = [shard_range(n_clips, r, world) for r in range(world)]
    bmax = max(hi - lo for lo, hi in sizes)
    if local.shape[0] < bmax:
        pad = torch.zeros(bmax - local.shape[0], *local.shape[1:], device=local.device, dtype=local.dtype)
        local = torch.cat([local, pad], 0)
    if dist.get_backend() == "gloo" and local.is_cuda:
        # test configuration only (several ranks sharing one GPU: RCCL refuses duplicate devices): gloo moves host memory
        host = torch.empty(world * bmax, *local.shape[1:], dtype=local.dtype)
        dist.all_gather_into_tensor(host, local.contiguous().cpu())
        out = host.to(local.device)
    else:
        out = torch.empty(world * bmax, *local.shape[1:], device=local.device, dtype=local.dtype)
        dist.all_gather_into_tensor(out, local.contiguous())
    out = out.view(world, bmax, *local.shape[1:])
    return torch.cat([out[r, : hi - lo] for r, (lo, hi) in enumerate(sizes)], 0)


def max_over_ranks(seconds: float, device) -> float:
    if not collectives_on():
        return seconds
    t = torch.tensor([seconds], device="cpu" if dist.get_backend() == "gloo" else device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if collectives_on():
        dist.barrier()


def long_ddim_sample_sharded(diff, n_windows: int, Lq: int, nfeat: int, cond_local: torch.Tensor, x0_local, *, seed: int,
                             init_noise: torch.Tensor | None = None, **kw):
    """`GaussianDiffusion.long_ddim_sample` over the windows of one song sharded in contiguous blocks: the coupling
    inside a rank is part of the captured step, the one boundary per rank pair crosses RCCL point-to-point after every
    step (tcdiff_amd/stitch.py halo_exchange), and the result is all-gathered (reference model/diffusion.py:446-515 has
    no multi-GPU path).

    Every random draw is a function of the GLOBAL window index, never of a rank's own generator: `seed` (required, the
    same on every rank) keys the per-step Philox noise by (seed, global window, step, element), and the initial x_T of ALL
    n_windows windows comes from one generator seeded with it (or from `init_noise` (n_windows, Lq, nfeat), identical on
    every rank), of which a rank takes its slice -- so the result equals the single-process one for any world size.  Ranks
    beyond the number of windows would own nothing and break the halo chain: n_windows >= world is required."""
    from .stitch import halo_exchange
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    if n_windows < world:
        raise ValueError(f"long_ddim_sample_sharded: {n_windows} windows cannot be sharded over {world} ranks "
                         "(every rank must own at least one window)")
    if seed is None:
        raise ValueError("long_ddim_sample_sharded needs an explicit seed, identical on every rank")
    lo, hi = shard_range(n_windows, rank, world)
    if init_noise is None:
        g = torch.Generator().manual_seed(int(seed))
        init_noise = torch.randn(n_windows, Lq, nfeat, generator=g)
    elif tuple(init_noise.shape) != (n_windows, Lq, nfeat):
        raise ValueError("init_noise must hold x_T of ALL windows: (n_windows, Lq, nfeat)")
    row = (Lq // diff.seq_len) * nfeat
    out = diff.long_ddim_sample((hi - lo, Lq, nfeat), cond_local, x0_local, clip_offset=lo, seed=int(seed),
                                init_noise=init_noise[lo:hi],
                                halo_exchange=(lambda xv: halo_exchange(xv, diff.seq_len, row)) if world > 1 else None,
                                **kw)
    return gather_samples(out, n_windows)


class FlatGradientAllReducer:
    """Gradient averaging of the HIP training step (tcdiff_amd/train_engine.py) across the ranks of the default process
    group: the reference's intent of `accelerate launch` + DDP (TCDiff.py:51-52,108-111,232).

    The training engine writes every parameter gradient into ONE flat fp32 buffer, laid out so that a decoder layer's
    linears are contiguous; `ready(flat, lo, hi)` is called by the backward pass as soon as such a range is complete
    (layer 7 first) and launches its all-reduce asynchronously -- on RCCL's own stream, behind an event of the compute
    stream -- so the collective of layer l runs under the backward kernels of layers l-1 .. 0.  No copies into buckets: the
    collective reads and writes the gradient memory itself.  xGMI is point-to-point (7 links x ~153 GB/s per GPU), a ring
    all-reduce is bound by one link whatever the message size, so ranges are sent whole (13.6 MB per decoder layer; the
    tail of ~100 MB in `bucket_bytes` pieces) rather than in NVSwitch-style 25 MB buckets.
    `finish()` makes the compute stream wait for all of them; with RCCL the averaging is the collective's own
    ReduceOp.AVG, with gloo (CPU tests) the sum is scaled afterwards."""

    def __init__(self, bucket_bytes: int = 64 << 20, group=None):
        self.bucket = max(1, bucket_bytes // 4)
        self.group = group            # None: the default process group; bound by the trainer (TrainEngine.enable_grad_sync)
        self.work = []
        self.launched = 0

    def active(self) -> bool:
        return collectives_on(self.group)

    def ready(self, flat: torch.Tensor, lo: int, hi: int, more: bool = True):
        """`more` (whether the backward has launches left behind this range) only matters to the capture stand-in of
        train_engine._capture_bwd_segments, which shares this signature"""
        if not self.active() or hi <= lo:
            return
        avg = dist.get_backend(self.group) == "nccl"
        for a in range(lo, hi, self.bucket):
            piece = flat[a:min(hi, a + self.bucket)]
            w = dist.all_reduce(piece, op=dist.ReduceOp.AVG if avg else dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.work.append((w, None if avg else piece))
            self.launched += 1

    def finish(self) -> int:
        world = dist.get_world_size(self.group) if self.active() else 1
        n = len(self.work)
        for w, piece in self.work:
            w.wait()
            if piece is not None:
                piece.div_(world)           # gloo has no ReduceOp.AVG (test backend; RCCL averages in the collective)
        self.work = []
        return n
