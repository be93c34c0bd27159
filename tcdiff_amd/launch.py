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
    exit_code is 0 only if every rank exited 0."""
    if world < 1:
        raise ValueError("world must be >= 1")
    port = free_port()
    procs = []
    for r in range(world):
        env = rank_env(r, world, port)
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen([python, *argv], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    deadline = None if timeout is None else time.time() + timeout
    outs: List[Tuple[str, str]] = [("", "")] * world
    rc = 0
    for r, p in enumerate(procs):
        try:
            left = None if deadline is None else max(1.0, deadline - time.time())
            o, e = p.communicate(timeout=left)
        except subprocess.TimeoutExpired:
            for q in procs:            # exact PIDs we started, nothing pattern-based
                if q.poll() is None:
                    q.kill()
            o, e = p.communicate()
            rc = rc or 124
        outs[r] = (o, e)
        if p.returncode != 0:
            rc = rc or (p.returncode if p.returncode is not None else 1)
    return rc, outs[0][0], [e[-2000:] for _, e in outs]
