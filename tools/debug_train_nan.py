"""Diagnostic: run one training step and report the first backward stage whose outputs are not finite."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import tcdiff_oracle as O
from tcdiff_amd import kernels as K
import tcdiff_amd.train_engine as TE
from tcdiff_amd.model import DanceDecoder
from tcdiff_amd.diffusion import GaussianDiffusion

compute = sys.argv[1] if len(sys.argv) > 1 else "f32"
dn, S, T, b = 2, 60, 100, 3
sd = O.synth_state_dict(dn=dn, seq_len=S)
model = DanceDecoder(nfeats=151, seq_len=S, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                     cond_feature_dim=438, activation=F.gelu, required_dancer_num=dn, compute_dtype=compute)
model.load_state_dict(sd)
diff = GaussianDiffusion(model, S, 151, None, schedule="cosine", n_timestep=T, predict_epsilon=False, loss_type="l2",
                         cond_drop_prob=0.25, guidance_weight=2, seq_len=S).to("cuda")
diff.eval()
TE.TrainEngine.poison = True
bad = []

def fin(name, *ts):
    for i, t in enumerate(ts):
        if t is not None and torch.is_tensor(t) and t.is_floating_point() and not bool(torch.isfinite(t.float()).all()):
            bad.append(name)
            print("NON-FINITE:", name, i, tuple(t.shape), "nan frac", float(torch.isnan(t.float()).float().mean()))
            return

ob = TE._Lin.bwd
def lin_bwd(self, dY, ld, M, Xs, want):
    fin(f"lin {self.key} in dY", dY)
    r = ob(self, dY, ld, M, Xs, want)
    for w in want:
        if w is not None:
            fin(f"lin {self.key} dX", w[1] if w[0] != "HEADS" else w[1]["out"])
    fin(f"lin {self.key} gW", self.eng.gW[self.key])
    return r
TE._Lin.bwd = lin_bwd
orb = TE.TrainEngine.row_bwd
def row_bwd(self, **kw):
    r = orb(self, **kw)
    fin(f"row_bwd ln={kw.get('ln')} nln={kw.get('nln')} d_z", kw.get("d_z"))
    fin(f"row_bwd ln={kw.get('ln')} nln={kw.get('nln')} d_xres", kw.get("d_xres"))
    return r
TE.TrainEngine.row_bwd = row_bwd
oab = K.attention_bwd
def att_bwd(dt, Q, Kk, V, O_, dO, lse, delta, dQ, *a):
    fin("attention_bwd in dO", dO); fin("attention_bwd in lse", lse)
    r = oab(dt, Q, Kk, V, O_, dO, lse, delta, dQ, *a)
    fin("attention_bwd delta", delta); fin("attention_bwd dQ", dQ)
    return r
K.attention_bwd = att_bwd
oact = TE.TrainEngine.act_bwd
def act_bwd(self, a, dy, *r, **k):
    out = oact(self, a, dy, *r, **k)
    fin("act_bwd", out)
    return out
TE.TrainEngine.act_bwd = act_bwd

x_start = torch.stack([O.synth_motion(c, dn * S).reshape(S, dn, 151).permute(1, 0, 2) for c in range(b)])
cond = torch.stack([O.synth_cond(c, S) for c in range(b)])
noise = torch.stack([O.synth_xT(10 + c, dn * S).reshape(S, dn, 151) for c in range(b)])
t = torch.tensor([73, 5, 40]); keep = torch.tensor([True, False, True])
total, losses = diff.p_losses(x_start.cuda(), cond.cuda(), t.cuda(), noise=noise.cuda(), keep_mask=keep.cuda())
print("total", float(total))
total.backward()
print("first offenders:", bad[:5])
g = model.final_layer.weight.grad
print("final_layer.weight grad finite:", bool(torch.isfinite(g).all()), float(g.norm()))
