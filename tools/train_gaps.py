"""Diagnostic: where is the GPU idle inside a steady-state training step?  torch.profiler over one step: the device timeline's idle gaps
(> 3 us) with the kernels on either side, and the sum of kernel time against the step's wall time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.train_bench import build, train_step_fn
from tcdiff_amd import Adan
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model, diff = build("bf16")
optim = Adan(model.parameters(), lr=2e-4, weight_decay=0.02)
x = torch.randn(B, 3, 150, 151, device="cuda")
cond = torch.randn(B, 301, 438, device="cuda")
step = train_step_fn(diff, optim, x, cond)
for _ in range(6):
    step()
torch.cuda.synchronize()
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
    for _ in range(3):
        step()
    torch.cuda.synchronize()
ev = sorted(((e.time_range.start, e.time_range.end, e.name) for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA),
            key=lambda t: t[0])
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
print(f"3 steps: device span {span / 1e3:.2f} ms, kernels {busy / 1e3:.2f} ms, idle {(span - busy) / 1e3:.2f} ms over {len(ev)} launches")
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    if s1 - e0 > 3:
        gaps.append((s1 - e0, n0[:60], n1[:60]))
gaps.sort(reverse=True)
print(f"{len(gaps)} gaps > 3 us, total {sum(g for g, _, _ in gaps) / 1e3:.2f} ms; the largest:")
for g, a, b in gaps[:25]:
    print(f"  {g:8.1f} us   after {a}   before {b}")
small = sum(s1 - e0 for (s0, e0, _), (s1, e1, _) in zip(ev, ev[1:]) if 0 < s1 - e0 <= 3)
print(f"gaps <= 3 us: {small / 1e3:.2f} ms")
