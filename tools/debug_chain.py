import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcdiff_amd import _lib as L, kernels as K
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_kernels_gpu as T
DEV = "cuda"
def run(d, M, Lq):
    want = T._tail_unfused(d, M, Lq, False)
    bf = torch.bfloat16
    xo = torch.zeros(M, 512, device=DEV); ho = torch.zeros(M, 512, device=DEV, dtype=bf); ro = torch.zeros(M, 512, device=DEV, dtype=bf)
    fld = d["film"].shape[1]
    K.chain_tail(d["O"], d["Wfc"], d["lnp_g"], d["lnp_b"], d["film"][:, 1024:], d["film"][:, 2048:], fld, d["xres"], d["ln3_g"],
                 d["ln3_b"], d["W1"], d["b1"], d["W2"], d["b2"], d["ln4_g"], d["ln4_b"], d["W3"], d["b3"], xo, d["ln1_g"], d["ln1_b"], ho, ro, d["rope"], M, Lq)
    torch.cuda.synchronize()
    err = (xo - want[0]).abs()
    return T.relerr(xo, want[0]), err
M, Lq = 128, 64
base = T._tail_inputs(M, Lq)
print("full:", run(base, M, Lq)[0])
d = dict(base); d["W2"] = torch.zeros_like(base["W2"]); print("W2=0 (fc + lin3 only):", run(d, M, Lq)[0])
d = dict(base); d["W1"] = torch.zeros_like(base["W1"]); print("W1=0 (h1 = gelu(b1) const):", run(d, M, Lq)[0])
d = dict(base); d["Wfc"] = torch.zeros_like(base["Wfc"]); print("Wfc=0:", run(d, M, Lq)[0])
# localise: which hidden chunk of W2 matters
for c in range(8):
    d = dict(base); w2 = torch.zeros_like(base["W2"]); w2[:, c*128:(c+1)*128] = base["W2"][:, c*128:(c+1)*128]; d["W2"] = w2
    e, err = run(d, M, Lq)
    print(f"only hidden chunk {c}: rel {e:.3e}; worst rows {err.max(1).values.topk(3).indices.tolist()} worst cols {err.max(0).values.topk(3).indices.tolist()}")
e, err = run(base, M, Lq)
print("row error profile (max per row) first 70:", [round(float(v), 3) for v in err.max(1).values[:70]])
