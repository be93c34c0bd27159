"""Diagnostic: board power, shader clock and temperature while the sampler runs (rocm-smi polled every 0.2 s from a side
thread): is the benchmark at the power limit?  python tools/power_watch.py [seconds]"""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tcdiff_amd import DanceDecoder, GaussianDiffusion
from tcdiff_amd import weights as W

samples, stop = [], False
def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True,
                                 timeout=5).stdout
            d = json.loads(out)
            c = d[sorted(d)[0]]
            samples.append((time.time(), {k: v for k, v in c.items() if any(s in k.lower() for s in ("power", "sclk", "junction", "edge"))}))
        except Exception as e:
            samples.append((time.time(), {"error": repr(e)[:100]}))
        time.sleep(0.2)

dev = torch.device("cuda", 0)
model = DanceDecoder(nfeats=151, seq_len=150, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                     cond_feature_dim=438, activation=F.gelu, required_dancer_num=3, compute_dtype="bf16")
model.load_state_dict(W.synth_state_dict_like(model))
diff = GaussianDiffusion(model.eval(), 150, 151, None, schedule="cosine", n_timestep=1000, predict_epsilon=False, loss_type="l2",
                         use_p2=False, cond_drop_prob=0.25, guidance_weight=2, seq_len=150).to(dev).eval()
cond = torch.stack([W.synth_cond(c, 150) for c in range(16)]).to(dev)
xT = torch.stack([W.synth_xT(c, 450) for c in range(16)]).to(dev)
diff.p_sample_loop((16, 450, 151), cond, noise=xT, seed=1)
torch.cuda.synchronize()
th = threading.Thread(target=poll, daemon=True); th.start()
time.sleep(1.0)
t_idle = time.time()
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 12.0
t0 = time.time(); n = 0
while time.time() - t0 < secs:
    diff.p_sample_loop((16, 450, 151), cond, noise=xT, seed=1); torch.cuda.synchronize(); n += 1
t1 = time.time()
time.sleep(1.0)
stop = True; th.join()
print(f"{n} jobs in {t1 - t0:.2f} s = {16 * n / (t1 - t0):.2f} clips/s")
for t, d in samples:
    tag = "idle" if t < t_idle or t > t1 else "run "
    print(tag, f"{t - t0:6.2f}", d)
