"""One process per GPU: start N ranks of a script from a parent that never touches the GPU.

The reference samples on rank 0 only (TCDiff.py:251) and relies on `accelerate launch` for its process group
(TCDiff.py:51-54); here `bench.py --gpus N` (and any caller of `spawn_ranks`) starts the N ranks itself: the parent
imports nothing that initialises HIP, sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT for every
child, waits for all of them and relays rank 0's stdout.  Children are fresh interpreters (subprocess, never
`exec` after a GPU call); a child that fails makes the whole launch fail.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import time
from typing import Dict, List, Optional, Sequence, Tuple


def free_port() -> int:
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: required by RCCL on this pool
    return env


def spawn_ranks(argv: Sequence[str], world: int, *, python: str = sys.executable, timeout: Optional[float] = None,
                extra_env: Optional[Dict[str, str]] = None) -> Tuple[int, str, List[str]]:
    """Run `python argv...` as `world` ranks.  Returns (exit_code, rank-0 stdout, per-rank stderr tails).
    exit_code is 0 only if every rank exited 0.

    Every rank writes to its own temporary files (a pipe that nobody drains blocks the writer after ~64 KB -- RCCL debug
    logs, build output -- and with it the collective every other rank is waiting in), all ranks are polled together, and
    as soon as one exits non-zero (or the deadline passes) the others are killed by PID instead of being left to wait for
    a peer that will never arrive."""
    import tempfile
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs, files = [], []
    for r in range(world):
        env = rank_env(r, world, port)
        if extra_env:
            env.update(extra_env)
        fo, fe = tempfile.TemporaryFile("w+"), tempfile.TemporaryFile("w+")
        files.append((fo, fe))
        procs.append(subprocess.Popen([python, *argv], env=env, stdout=fo, stderr=fe, text=True))
    deadline = None if timeout is None else time.time() + timeout
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        failed = [c for c in codes if c not in (None, 0)]
        if failed:
            rc = failed[0]
            break
        if all(c == 0 for c in codes):
            break
        if deadline is not None and time.time() > deadline:
            rc = 124
            break
        time.sleep(0.05)
    for p in procs:                        # exact PIDs we started, nothing pattern-based
        if p.poll() is None:
            p.kill()
    for p in procs:
        p.wait()
    outs = []
    for fo, fe in files:
        fo.seek(0)
        fe.seek(0)
        outs.append((fo.read(), fe.read()))
        fo.close()
        fe.close()
    return rc, outs[0][0], [e[-2000:] for _, e in outs]
