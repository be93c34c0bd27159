"""Diagnostic: host-side time line of one steady-state one-clip ddim_sample job (perf_counter around the host stages; the device runs
asynchronously behind them)."""
import os, sys, time
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
marks = []
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        r = f(*a, **k)
        marks.append((name, t0, time.perf_counter()))
        return r
    setattr(obj, name, g)
for _ in range(4):
    diff.ddim_sample((nb, Lq, 151), cond, x_0=x0, init_noise=xT, seed=1)
torch.cuda.synchronize()
eng = model.engine(nb)
for n in ("_prepare", "_ddim_params", "_ddim_pairs", "_time_rows"):
    wrap(diff, n)
for n in ("encode_music", "fill_kv_slots", "build_film_table", "sampler_state"):
    wrap(eng, n)
import torch.cuda
orig_replay = torch.cuda.CUDAGraph.replay
def replay(self):
    t0 = time.perf_counter(); orig_replay(self); marks.append(("graph.replay", t0, time.perf_counter()))
torch.cuda.CUDAGraph.replay = replay
torch.cuda.synchronize()
T0 = time.perf_counter()
out = diff.ddim_sample((nb, Lq, 151), cond, x_0=x0, init_noise=xT, seed=1)
T1 = time.perf_counter()
torch.cuda.synchronize()
T2 = time.perf_counter()
print(f"ddim_sample returned after {(T1 - T0) * 1e3:.2f} ms of host time; device done at {(T2 - T0) * 1e3:.2f} ms")
for n, a, b in marks:
    print(f"  {n:18s} starts {(a - T0) * 1e3:7.3f} ms  takes {(b - a) * 1e3:7.3f} ms")
