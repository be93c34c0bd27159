"""world_size-2 gloo test of the clip sharding + result all-gather used by bench.py --gpus N (runs on CPU)."""
import os
import socket

import torch
import torch.multiprocessing as mp

from tcdiff_amd import dist as D


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _sample_of(clip: int, Lq=6, F=5):
    g = torch.Generator().manual_seed(clip)
    return torch.randn(Lq, F, generator=g)


def _worker(rank, world, port, n_clips, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    r, w, _ = D.init_from_env("gloo")
    lo, hi = D.shard_range(n_clips, r, w)
    local = torch.stack([_sample_of(c) for c in range(lo, hi)]) if hi > lo else torch.zeros(0, 6, 5)
    D.barrier()
    full = D.gather_samples(local, n_clips)
    t = D.max_over_ranks(1.0 + rank, "cpu")
    want = torch.stack([_sample_of(c) for c in range(n_clips)])
    q.put((rank, bool(torch.equal(full, want)), t, (lo, hi)))
    torch.distributed.destroy_process_group()


def _run(n_clips):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, n_clips, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_equal_shards_allgather_in_global_order():
    res = _run(8)
    assert [r[1] for r in res] == [True, True]
    assert [r[3] for r in res] == [(0, 4), (4, 8)]
    assert all(abs(r[2] - 2.0) < 1e-9 for r in res)       # max over ranks


def test_ragged_shards():
    res = _run(5)
    assert [r[1] for r in res] == [True, True]
    assert [r[3] for r in res] == [(0, 3), (3, 5)]


def test_shard_range_covers_everything():
    for n in (1, 7, 16, 128):
        for w in (1, 2, 4, 8):
            spans = [D.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _subgroup_worker(rank, world, port, q):
    """gradient averaging bound to an explicit process group: rank 2 of 3 is outside the group and must neither take part
    nor block (a default group that exists for sharded sampling must not pull a fine-tuning rank into collectives)"""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    D.init_from_env("gloo")
    grp = torch.distributed.new_group(ranks=[0, 1])
    flat = torch.full((5000,), float(rank + 1))
    if rank < 2:
        red = D.FlatGradientAllReducer(bucket_bytes=4096, group=grp)
        assert red.active()
        red.ready(flat, 0, 3000)
        red.ready(flat, 3000, 5000)
        n = red.finish()
    else:
        n = 0                                          # trains nothing, enters no collective
    q.put((rank, float(flat.min()), float(flat.max()), n))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_gradient_allreduce_is_bound_to_the_group_it_was_given():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_subgroup_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1:3] == (1.5, 1.5) and res[1][1:3] == (1.5, 1.5) and res[2][1:3] == (3.0, 3.0)
    assert res[0][3] == res[1][3] == 3 + 2 and res[2][3] == 0
