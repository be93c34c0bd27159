"""Data-parallel training on the real kernels: two ranks share the ONE GPU of the test box (gloo moves the gradient slices;
on a multi-GPU node the same code runs with backend nccl = RCCL, one rank per GPU).  What is checked is what the driver's
8-GPU run depends on: the backward launches its per-layer all-reduces on the flat gradient buffer, every rank ends each step
with identical parameters, the averaged gradient equals the single-process gradient of the concatenated batch, and the
replayed forward (hipGraph from the third step on) coexists with the eager, collective-launching backward."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ["TC_ROOT"])
import numpy as np, torch, torch.nn.functional as F
from oracle import tcdiff_oracle as O                      # synthetic weights / inputs only
from tcdiff_amd import Adan, dist as D
from tcdiff_amd.diffusion import GaussianDiffusion
from tcdiff_amd.model import DanceDecoder
rank, world, _ = D.init_from_env("gloo")
torch.cuda.set_device(0)
DEV, DN, S, T, b = "cuda", 2, 60, 100, 2
model = DanceDecoder(nfeats=151, seq_len=S, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                     cond_feature_dim=438, activation=F.gelu, required_dancer_num=DN, compute_dtype="f32")
model.load_state_dict(O.synth_state_dict(dn=DN, seq_len=S))
diff = GaussianDiffusion(model, S, 151, None, schedule="cosine", n_timestep=T, predict_epsilon=False, loss_type="l2",
                         use_p2=False, cond_drop_prob=0.25, guidance_weight=2, seq_len=S).to(DEV)
diff.eval()                                                # dropout off: the global batch has a single-process equivalent
optim = Adan(model.parameters(), lr=1e-4, weight_decay=0.02)
if world > 1:
    model.train_engine().enable_grad_sync()                # opt-in: bound to the default group by the trainer
n_steps = 5
def data(step, r):
    c0 = 10 * step + b * r
    x = torch.stack([O.synth_motion(c0 + c, DN * S).reshape(S, DN, 151).permute(1, 0, 2) for c in range(b)])
    cond = torch.stack([O.synth_cond(c0 + c, S) for c in range(b)])
    noise = torch.stack([O.synth_xT(c0 + c, DN * S).reshape(S, DN, 151) for c in range(b)])
    t = torch.tensor([(7 * (c0 + c) + 3) % T for c in range(b)])
    keep = torch.tensor([(c0 + c) % 3 != 0 for c in range(b)])
    return x, cond, noise, t, keep
out = {"rank": rank, "world": world}
ranks = range(world) if world > 1 else range(int(os.environ["TC_AS_WORLD"]))
for step in range(n_steps):
    if world > 1:
        parts = [data(step, rank)]
    else:                                                  # the reference run: every rank's clips in one batch
        parts = [data(step, r) for r in ranks]
    x, cond, noise, t, keep = (torch.cat([p[i] for p in parts]) for i in range(5))
    total, _ = diff.p_losses(x.to(DEV), cond.to(DEV), t.to(DEV), noise=noise.to(DEV), keep_mask=keep.to(DEV))
    optim.zero_grad()
    total.backward()
    if step == 0:                                          # identical parameters in every run: gradients comparable to fp32 ordering
        out["grad"] = {n: float(p.grad.double().norm()) for n, p in model.named_parameters() if p.grad is not None}
        out["gsample"] = model.final_layer.weight.grad.detach().flatten()[:64].cpu().tolist()
    optim.step()
eng = model.train_engine()
out["collectives"] = eng.grad_sync.launched if eng.grad_sync else 0
out["graph_broken"] = eng._graph_broken
out["fwd_replayed"] = any(st["fwd"] is not None for st in eng._graphs.values())
out["bwd_captured"] = any(st["bwd"] is not None for st in eng._graphs.values())
out["bwd_segments"] = max([len(st["bwd_segs"]) for st in eng._graphs.values() if st.get("bwd_segs")] or [0])
out["psum"] = {n: float(p.detach().double().sum()) for n, p in model.named_parameters()}
print("RESULT " + json.dumps(out), flush=True)
if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, as_world=2):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, TC_ROOT=ROOT, TC_AS_WORLD=str(as_world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        if world > 1:
            env.update(RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0")
        else:
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
                env.pop(k, None)
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    res = []
    for p in procs:
        try:
            so, se = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, se[-3000:]
        line = [ln for ln in so.splitlines() if ln.startswith("RESULT ")][-1]
        res.append(json.loads(line[7:]))
    return sorted(res, key=lambda d: d["rank"])


def test_two_ranks_average_gradients_and_stay_identical():
    two = _run(2)
    one = _run(1)[0]
    a, b = two
    assert a["world"] == b["world"] == 2 and one["world"] == 1
    assert a["collectives"] == b["collectives"] and a["collectives"] >= 5 * 10         # >= 10 all-reduces per step, same on both
    assert a["graph_broken"] is None and b["graph_broken"] is None
    # forward replayed from step 3 on; the data-parallel backward is replayed too, as one graph per decoder layer + the front,
    # with the layer's all-reduce launched between two replays (steps 3 and 4 of this run: capture + replay, then replay)
    assert a["fwd_replayed"] and not a["bwd_captured"] and a["bwd_segments"] == b["bwd_segments"] == 9
    assert one["bwd_captured"] and one["bwd_segments"] == 0   # without a process group: ONE backward graph
    # identical parameters on both ranks after 5 steps (same averaged gradients, deterministic fused Adan)
    worst = max(abs(a["psum"][n] - b["psum"][n]) for n in a["psum"])
    assert worst == 0.0, worst
    # step 0 (same parameters everywhere): the averaged gradient = the single-process gradient of the concatenated batch
    # (the loss is a mean over clips); later steps drift apart through Adan's moment ratios, as any two fp32 summation orders do
    rel = max(abs(a["grad"][n] - one["grad"][n]) / (one["grad"][n] + 1e-30) for n in one["grad"])
    gs = float(np.abs(np.array(a["gsample"]) - np.array(one["gsample"])).max() / (np.abs(np.array(one["gsample"])).max() + 1e-30))
    assert set(a["grad"]) == set(one["grad"]) and rel < 2e-5 and gs < 2e-5, (rel, gs)
