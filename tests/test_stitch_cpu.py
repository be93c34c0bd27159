"""Long mode: the cross-rank boundary copy (world_size-2 gloo) and the window stitch (reference
model/diffusion.py:502-506,841-897, dataset/quaternion.py:35-71)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp
from scipy.spatial.transform import Rotation as R, Slerp

from tcdiff_amd import dist as D
from tcdiff_amd import stitch as S


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _windows(n, L=12, F=5):
    g = torch.Generator().manual_seed(11)
    return torch.randn(n, L, F, generator=g)


def _worker(rank, world, port, n, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    D.init_from_env("gloo")
    seq_len, dn, F = 6, 2, 5                 # L = seq_len * dn tokens per window
    full = _windows(n)
    lo, hi = D.shard_range(n, rank, world)
    x = full[lo:hi].clone()
    for _ in range(3):                       # three "steps": local coupling, then the cross-rank boundary
        xv = x.view(hi - lo, seq_len, dn * F)
        if hi - lo > 1:
            xv[1:, :seq_len // 2] = xv[:-1, seq_len // 2:].clone()
        S.halo_exchange(x, seq_len, dn * F)
        x = x * 1.5 + 0.25                   # stand-in for the next denoising step
    got = D.gather_samples(x, n)
    want = full.clone()
    for _ in range(3):
        wv = want.view(n, seq_len, dn * F)
        wv[1:, :seq_len // 2] = wv[:-1, seq_len // 2:].clone()
        want = want * 1.5 + 0.25
    q.put((rank, bool(torch.equal(got, want))))
    torch.distributed.destroy_process_group()


def test_halo_exchange_equals_single_process_coupling():
    for n in (4, 5, 2):                      # equal shards, ragged shards, one window per rank
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = _free_port()
        ps = [ctx.Process(target=_worker, args=(r, 2, port, n, q)) for r in range(2)]
        for p in ps:
            p.start()
        res = sorted(q.get(timeout=120) for _ in ps)
        for p in ps:
            p.join(timeout=60)
            assert p.exitcode == 0
        assert [r[1] for r in res] == [True, True], n


def test_stitch_of_consistent_windows_returns_the_sequence():
    """windows cut from ONE sequence (so the overlaps agree) stitch back to it: fades sum to 1, slerp of equal ends"""
    g = torch.Generator().manual_seed(3)
    s, half, b, dn = 8, 4, 4, 2
    total = s + half * (b - 1)
    pos_seq = torch.randn(total, dn, 3, generator=g, dtype=torch.float64)
    q_seq = torch.randn(total, dn, 24, 3, generator=g, dtype=torch.float64) * 0.6
    pos = torch.stack([pos_seq[i * half:i * half + s] for i in range(b)])
    q = torch.stack([q_seq[i * half:i * half + s] for i in range(b)])
    fp, fq = S.stitch_windows(pos, q)
    assert fp.shape == (total, dn, 3) and fq.shape == (total, dn, 24, 3)
    # interior frames: fade_out + fade_in = linspace(1,0) + linspace(0,1) = 1
    assert float((fp - pos_seq).abs().max()) < 1e-12
    assert float((fq - q_seq).abs().max()) < 1e-9


def test_slerp_against_scipy():
    g = torch.Generator().manual_seed(4)
    a = torch.randn(6, 3, generator=g, dtype=torch.float64)
    b = a + 0.8 * torch.randn(6, 3, generator=g, dtype=torch.float64)
    qa, qb = S._axis_angle_to_quaternion(a), S._axis_angle_to_quaternion(b)
    for w in (0.0, 0.3, 0.9):
        got = S.quat_slerp(qa, qb, torch.tensor(w, dtype=torch.float64))
        for i in range(6):
            key = R.from_rotvec(np.stack([a[i].numpy(), b[i].numpy()]))
            want = Slerp([0, 1], key)([w]).as_matrix()[0]
            gq = got[i].numpy()
            have = R.from_quat([gq[1], gq[2], gq[3], gq[0]]).as_matrix()
            linear = 1.0 - abs(float((qa[i] * qb[i]).sum())) < 0.01     # the reference lerps (un-normalised) when close
            assert np.abs(have - want).max() < (2e-4 if linear else 1e-9)


# ---- dist.long_ddim_sample_sharded: every draw is keyed by the GLOBAL window index ------------------------------------
class _FakeLongSampler:
    """Stands in for GaussianDiffusion.long_ddim_sample on CPU (the HIP sampler needs a GPU): a few "steps" of
    x <- 0.5 x + noise(seed, global window, step), the in-rank window coupling, then the cross-rank halo hook -- the
    contract dist.long_ddim_sample_sharded relies on (init_noise, seed, clip_offset, halo_exchange)."""
    seq_len = 6

    def long_ddim_sample(self, shape, cond, x_0, *, clip_offset, seed, init_noise, halo_exchange=None):
        b, L, F = shape
        assert init_noise.shape == shape and cond.shape[0] == b
        x = init_noise.clone()
        half = self.seq_len // 2
        for step in range(4):
            for w in range(b):
                g = torch.Generator().manual_seed(seed * 1000003 + (clip_offset + w) * 101 + step)
                x[w] = 0.5 * x[w] + torch.randn(L, F, generator=g)
            xv = x.view(b, self.seq_len, -1)
            if b > 1:
                xv[1:, :half] = xv[:-1, half:].clone()
            if halo_exchange is not None:
                halo_exchange(x)
        return x


def _sharded_worker(rank, world, port, n, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    D.init_from_env("gloo")
    torch.manual_seed(1234)                  # the usual DDP setup: the SAME global seed on every rank
    lo, hi = D.shard_range(n, rank, world)
    cond = torch.zeros(hi - lo, 3, 4)
    out = D.long_ddim_sample_sharded(_FakeLongSampler(), n, 12, 5, cond, None, seed=77)
    try:
        D.long_ddim_sample_sharded(_FakeLongSampler(), 1, 12, 5, cond[:1], None, seed=77)
        refused = False
    except ValueError:
        refused = True
    q.put((rank, out, refused))
    torch.distributed.destroy_process_group()


def test_sharded_long_sampler_equals_the_single_process_one():
    n = 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_sharded_worker, args=(r, 2, port, n, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted((q.get(timeout=120) for _ in ps), key=lambda t: t[0])
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, all windows: the same seed-derived x_T and per-window noise
    g = torch.Generator().manual_seed(77)
    init = torch.randn(n, 12, 5, generator=g)
    want = _FakeLongSampler().long_ddim_sample((n, 12, 5), torch.zeros(n, 3, 4), None, clip_offset=0, seed=77, init_noise=init)
    for rank, out, refused in res:
        assert torch.equal(out, want), rank            # identical on both ranks and to the unsharded run
        assert refused                                   # fewer windows than ranks is an error, not a hang
    assert not torch.equal(want[0], want[3])             # windows are not copies of one another
