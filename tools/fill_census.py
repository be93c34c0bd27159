"""Where do the training step's fill / copy launches come from?  One steady-state step with torch's fill and copy entry points
wrapped to record their Python call sites.  python tools/fill_census.py [--batch 32]"""
import argparse
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from tools.train_bench import build, train_step_fn
from tcdiff_amd import Adan

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
a = ap.parse_args()
model, diff = build("bf16")
optim = Adan(model.parameters(), lr=2e-4, weight_decay=0.02)
x = torch.randn(a.batch, 3, 150, 151, device="cuda")
cond = torch.randn(a.batch, 301, 438, device="cuda")
step = train_step_fn(diff, optim, x, cond)
for _ in range(5):
    step()
torch.cuda.synchronize()
sites = collections.Counter()


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*args, **kw):
        st = traceback.extract_stack(limit=6)[:-1]
        key = name + " <- " + " <- ".join(f"{os.path.basename(s.filename)}:{s.lineno}" for s in reversed(st) if "fill_census" not in s.filename)
        sites[key] += 1
        return orig(*args, **kw)
    setattr(owner, name, f)


for owner, names in ((torch.Tensor, ["zero_", "fill_", "copy_", "clone", "contiguous", "to", "float"]),
                     (torch, ["zeros", "zeros_like", "full", "ones", "empty_like", "cat", "stack", "tensor", "as_tensor"])):
    for n in names:
        wrap(owner, n)
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
    step()
    torch.cuda.synchronize()
for k, v in sites.most_common(60):
    print(f"{v:5d}  {k}")
print("---- device kernels of the step that are not tcdiff's")
agg = collections.Counter()
tim = collections.Counter()
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA and not any(s in e.name for s in ("_kernel<", "_kernel(", "tcdiff")):
        agg[e.name[:110]] += 1
        tim[e.name[:110]] += e.device_time
for k, v in agg.most_common(25):
    print(f"{v:5d} {tim[k]:9.1f} us  {k}")
