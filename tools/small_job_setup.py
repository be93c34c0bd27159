"""Diagnostic: what a steady-state one-clip ddim_sample job spends OUTSIDE its 50 steps (per-job setup: music branch, tables, caches,
host work).  torch.profiler over the third job: device kernels that run fewer than 50 times, and wall time against device time."""
import collections, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tcdiff_amd import DanceDecoder, GaussianDiffusion
from tcdiff_amd import weights as W
dn, S, T, nb = 3, 150, 1000, 1
dev = torch.device("cuda", 0)
model = DanceDecoder(nfeats=151, seq_len=S, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1, cond_feature_dim=438,
                     activation=F.gelu, required_dancer_num=dn, compute_dtype="bf16")
model.load_state_dict(W.synth_state_dict_like(model))
diff = GaussianDiffusion(model.eval(), S, 151, None, schedule="cosine", n_timestep=T, predict_epsilon=False, loss_type="l2", use_p2=False,
                         cond_drop_prob=0.25, guidance_weight=2, seq_len=S).to(dev).eval()
Lq = dn * S
cond = torch.stack([W.synth_cond(c, S) for c in range(nb)]).to(dev)
xT = torch.stack([W.synth_xT(c, Lq) for c in range(nb)]).to(dev)
x0 = torch.stack([W.synth_xT(100 + c, Lq, 3) for c in range(nb)]).clamp(-1, 1).to(dev)
def job():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    diff.ddim_sample((nb, Lq, 151), cond, x_0=x0, init_noise=xT, seed=1)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
for i in range(4):
    print(f"job {i}: {job():.2f} ms")
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA, torch.profiler.ProfilerActivity.CPU]) as prof:
    ms = job()
print(f"profiled job: {ms:.2f} ms wall")
agg, tim = collections.Counter(), collections.Counter()
first, last = None, None
for e in prof.events():
    if e.device_type == torch.autograd.DeviceType.CUDA:
        agg[e.name[:100]] += 1
        tim[e.name[:100]] += e.device_time
        s, t = e.time_range.start, e.time_range.end
        first = s if first is None else min(first, s)
        last = t if last is None else max(last, t)
tot = sum(tim.values())
print(f"device kernels: {tot / 1e3:.2f} ms busy, first to last {((last - first) / 1e3):.2f} ms")
print("---- kernels with fewer than 50 launches in the job (setup), by time")
setup = 0.0
for k, v in sorted(tim.items(), key=lambda kv: -kv[1]):
    if agg[k] < 50:
        setup += v
        print(f"{agg[k]:5d} {v:9.1f} us  {k}")
print(f"setup kernels: {setup / 1e3:.3f} ms")
# host-side view: top CPU ops by self time
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=18, max_name_column_width=60))
