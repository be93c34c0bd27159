"""bench.py itself with --gpus 2 on the test box's ONE GPU (TCDIFF_BENCH_ONE_DEVICE=1: both ranks on cuda:0, gloo for the
collectives): the rank plumbing the driver's 8-GPU scaling run depends on -- self-spawned ranks, contiguous disjoint clip
ranges, global clip offsets for the noise, one gather, max-over-ranks timing, ONE JSON line from rank 0 -- and the samples of
the two-rank job equal, bit for bit, those of the one-rank job over the same global clips."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--steps", "1", "--warmup", "1", "--ddpm-steps", "24", "--no-cpu-baseline", "--no-parity-mode", "--no-kernel-profile",
         "--no-train-step", "--no-other-configs"]


def _bench(gpus, batch, dump, extra_env=None):
    env = dict(os.environ, TCDIFF_BENCH_ONE_DEVICE="1", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--batch", str(batch),
                        "--dump-samples", dump] + FLAGS, capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                     # exactly one JSON line, from rank 0
    return json.loads(lines[0])


def test_two_ranks_of_bench_py_on_one_gpu_equal_the_one_rank_job(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    one, two = str(tmp_path / "one.pt"), str(tmp_path / "two.pt")
    j1 = _bench(1, 4, one)
    j2 = _bench(2, 2, two)
    assert j1["n_gpus"] == 1 and j1["ranks_seen"] == 1 and j1["clip_ranges"] == [[0, 4]]
    assert j2["n_gpus"] == 2 and j2["ranks_seen"] == 2 and j2["clip_ranges"] == [[0, 2], [2, 4]]
    assert j2["scaling"] == "weak" and j2["config"]["clips_per_gpu"] == 2 and j2["value"] > 0
    a, b = torch.load(one), torch.load(two)
    assert a.shape == b.shape == (4, 450, 151)
    assert torch.equal(a, b), "a clip's sample depends on the rank layout"
