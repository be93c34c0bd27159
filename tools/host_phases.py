"""Host time of the phases of a training step (no syncs inside): where the host blocks.  python tools/host_phases.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.train_bench import build
from tcdiff_amd import Adan
b = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model, diff = build("bf16")
optim = Adan(model.parameters(), lr=2e-4, weight_decay=0.02)
x = torch.rand(b, 3, 150, 151, device="cuda") * 2 - 1
cond = torch.randn(b, 301, 438, device="cuda")
acc = [0.0] * 6
N = 20
for it in range(N + 6):
    if it == 6:
        torch.cuda.synchronize(); acc = [0.0] * 6; t_all = time.perf_counter()
    t0 = time.perf_counter()
    total, _ = diff(x, cond)
    t1 = time.perf_counter()
    optim.zero_grad()
    t2 = time.perf_counter()
    total.backward()
    t3 = time.perf_counter()
    optim.step()
    t4 = time.perf_counter()
    diff.ema.update_model_average(diff.master_model, diff.model)
    t5 = time.perf_counter()
    for i, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
        acc[i] += d
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) / N * 1e3
names = ["forward+loss", "zero_grad", "backward", "optim.step", "ema"]
print(f"batch {b} mode {os.environ.get('TCDIFF_TRAIN_GRAPH', '1')}: wall {wall:.2f} ms/step; host: " +
      ", ".join(f"{n} {a / N * 1e3:.2f}" for n, a in zip(names, acc)) + f"; host sum {sum(acc) / N * 1e3:.2f}")
