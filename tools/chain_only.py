"""run chain_tail (and the unfused equivalent) a few times -- target for rocprofv3 --pmc"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from tcdiff_amd import _lib as L, kernels as K
import test_kernels_gpu as T
M, Lq = 14400, 450
d = T._tail_inputs(M, Lq)
bf = torch.bfloat16
xo = torch.zeros(M, 512, device="cuda"); ho = torch.zeros(M, 512, device="cuda", dtype=bf); ro = torch.zeros(M, 512, device="cuda", dtype=bf)
fld = d["film"].shape[1]
for _ in range(5):
    K.chain_tail(d["O"], d["Wfc"], d["lnp_g"], d["lnp_b"], d["film"][:, 1024:], d["film"][:, 2048:], fld, d["xres"], d["ln3_g"],
                 d["ln3_b"], d["W1"], d["b1"], d["W2"], d["b2"], d["ln4_g"], d["ln4_b"], d["W3"], d["b3"], xo, d["ln1_g"], d["ln1_b"], ho, ro, d["rope"], M, Lq)
    T._tail_unfused(d, M, Lq, False)
torch.cuda.synchronize()
