"""world_size-2 gloo tests (CPU) of the training step's data-parallel gradient averaging (tcdiff_amd/dist.py
FlatGradientAllReducer driven by tcdiff_amd/train_engine.py's backward):

  * the reducer averages rank-specific gradients in place, identically on both ranks, range by range;
  * the training engine's backward issues the SAME sequence of collectives on every rank (a mismatch deadlocks) -- run here
    as a host-side dry run against a stub library (no GPU in this container; the kernels compute nothing, the collective
    schedule and the flat-buffer plumbing are what is exercised);
  * identical averaged gradients + the deterministic fused Adan give identical parameters on both ranks, shown with the
    optimizer's update rule evaluated by the CPU oracle on each rank's averaged gradients.
The numerics of the step itself are GPU tests (tests/test_train_step_gpu.py)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from tcdiff_amd import dist as D
    D.init_from_env("gloo")
    return D


def _reducer_worker(rank, world, port, q):
    D = _init(rank, world, port)
    n = 100_003
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    mine = flat.clone()
    red = D.FlatGradientAllReducer(bucket_bytes=64 * 1024)          # 16 k-element pieces: several collectives per range
    ranges = [(60_000, 80_000), (30_000, 60_000), (0, 30_000), (80_000, n)]      # the order a backward pass completes them in
    for lo, hi in ranges:
        red.ready(flat, lo, hi)
    n_coll = red.finish()
    others = [torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    want = sum(others) / world
    # the update rule on the averaged gradient (oracle's restatement of model/adan.py): same inputs -> same parameters
    from oracle import tcdiff_oracle as O
    p0 = np.linspace(-1, 1, 1000, dtype=np.float32)
    st = dict(step=1, m=np.zeros_like(p0), v=np.zeros_like(p0), n=np.zeros_like(p0), prev_grad=np.zeros_like(p0))
    p1 = O.adan_step(p0.copy(), flat[:1000].numpy().copy(), st, lr=5e-5, weight_decay=0.02)
    q.put((rank, float((flat - want).abs().max()), n_coll, bool(torch.equal(mine, flat)), p1.tobytes()))
    torch.distributed.destroy_process_group()


def _engine_worker(rank, world, port, q):
    D = _init(rank, world, port)
    import torch.nn.functional as F
    from tcdiff_amd import _lib as L, kernels as K
    import tcdiff_amd.train_engine as TE

    class _Stub:
        def __getattr__(self, name):
            return lambda *a: 0
    L._lib = _Stub()
    K.stream = lambda: 0
    TE._ALLOW_CPU = True
    from tcdiff_amd.model import DanceDecoder
    torch.manual_seed(0)
    model = DanceDecoder(nfeats=151, seq_len=8, latent_dim=512, ff_size=1024, num_layers=2, num_heads=8, dropout=0.1,
                         cond_feature_dim=438, activation=F.gelu, required_dancer_num=2, compute_dtype="bf16")
    eng = model.train_engine()
    assert eng.grad_sync is None              # a process group alone does NOT switch gradient averaging on (opt-in)
    eng.enable_grad_sync(group=torch.distributed.group.WORLD)
    calls = []
    orig_ready = D.FlatGradientAllReducer.ready

    def ready(self, flat, lo, hi, more=True):
        # the kernels are stubs: stand in for them with a rank-specific gradient in the range being published
        flat[lo:hi] = float(rank + 1)
        calls.append((lo, hi))
        return orig_ready(self, flat, lo, hi, more)
    D.FlatGradientAllReducer.ready = ready
    b = 2
    x, cond = torch.randn(b, 16, 151), torch.randn(b, 17, 438)
    out = TE.denoiser_train(model, x, cond, torch.tensor([3, 4]), torch.tensor([True, False]), (1, 2), 0.1)
    out.sum().backward()
    covered = sorted(calls)
    contiguous = covered[0][0] == 0 and covered[-1][1] == eng.n_grad and all(a[1] == b_[0] for a, b_ in zip(covered, covered[1:]))
    g = model.final_layer.weight.grad
    q.put((rank, calls, contiguous, float(g.min()), float(g.max())))
    torch.distributed.destroy_process_group()


def _run(worker):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=300) for _ in ps)
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_flat_reducer_averages_in_place_identically_on_both_ranks():
    res = _run(_reducer_worker)
    for rank, err, n_coll, unchanged, _ in res:
        assert err < 1e-6 and not unchanged
        assert n_coll == 2 + 2 + 2 + 2          # ceil(range / 16384) pieces per range
    assert res[0][4] == res[1][4]               # identical averaged gradients -> bit-identical parameters after the update


def test_training_backward_issues_the_same_collectives_on_every_rank():
    res = _run(_engine_worker)
    (r0, calls0, cont0, lo0, hi0), (r1, calls1, cont1, lo1, hi1) = res
    assert calls0 == calls1 and cont0 and cont1
    # decoder layers are published last-to-first while the backward is still running, the rest at the end
    assert len(calls0) == 2 + 2 and calls0[0][0] > calls0[1][0]
    # (1 + 2) / 2: every gradient element went through the average exactly once
    assert lo0 == hi0 == lo1 == hi1 == 1.5
