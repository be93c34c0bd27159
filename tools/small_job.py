"""Diagnostic: what TCDiff.py runs when it renders -- ddim_sample of ONE 3 x 150 clip, 50 steps -- three times (for a rocprofv3
--kernel-trace --stats wrapper): python tools/small_job.py [clips]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tcdiff_amd import DanceDecoder, GaussianDiffusion
from tcdiff_amd import weights as W
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dn, S, T = 3, 150, 1000
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
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    diff.ddim_sample((nb, Lq, 151), cond, x_0=x0, init_noise=xT, seed=1)
    torch.cuda.synchronize(); print(f"job {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms")
