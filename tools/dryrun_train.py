"""Host-side dry run of the training schedule (no GPU): the C-ABI library is replaced by a stub whose launchers return 0,
so that every tensor shape, keyword and pointer the Python schedule of tcdiff_amd/train_engine.py builds is exercised on CPU
tensors.  Catches host bugs before a GPU visit; computes nothing."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tcdiff_amd import _lib as L, kernels as K
import tcdiff_amd.train_engine as TE


class _Stub:
    def __getattr__(self, name):
        def fn(*a):
            return 0
        return fn


L._lib = _Stub()
K.stream = lambda: 0
TE._ALLOW_CPU = True
TE.TrainEngine.use_graphs = False      # no device, no capture
from tcdiff_amd.model import DanceDecoder
from tcdiff_amd.diffusion import GaussianDiffusion

dn, S, T, b = 2, 60, 100, 3
model = DanceDecoder(nfeats=151, seq_len=S, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                     cond_feature_dim=438, activation=F.gelu, required_dancer_num=dn, compute_dtype=sys.argv[1] if len(sys.argv) > 1 else "bf16")
model.train()
eng = TE.TrainEngine(model, model.compute_dtype)
model._train_engine = eng
x = torch.randn(b, dn * S, 151)
cond = torch.randn(b, 2 * S + 1, 438)
t = torch.randint(0, T, (b,))
keep = torch.tensor([True, False, True])
out = TE.denoiser_train(model, x, cond, t, keep, (1, 2), 0.1)
print("forward ok", out.shape, out.requires_grad)
out.sum().backward()
n_grad = sum(p.grad is not None for p in model.parameters())
print("backward ok; parameters with grad:", n_grad, "of", len(list(model.parameters())), "flat", eng.n_grad)
shapes_ok = all(p.grad.shape == p.shape for p in model.parameters() if p.grad is not None)
print("grad shapes ok:", shapes_ok)

# host cost of one forward + backward of the schedule (stub launchers: pure Python / ctypes / allocator time)
import time
import cProfile, pstats
for p in model.parameters():
    p.grad = None
t0 = time.perf_counter()
N = 5
for _ in range(N):
    out = TE.denoiser_train(model, x, cond, t, keep, (1, 2), 0.1)
    out.sum().backward()
    for p in model.parameters():
        p.grad = None
print(f"host time per forward + backward: {(time.perf_counter() - t0) / N * 1e3:.2f} ms")
if "--profile" in sys.argv:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(3):
        out = TE.denoiser_train(model, x, cond, t, keep, (1, 2), 0.1)
        out.sum().backward()
        for p in model.parameters():
            p.grad = None
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
