"""diagnostic (TC_STAMP build): phases of chain_tail"""
import sys, os, math, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from tcdiff_amd import _lib as L, kernels as K
import test_kernels_gpu as T
from tools.microbench import ev
lib = L.load()
lib.tcdiff_debug_chain_stamp_buffer.argtypes = [ctypes.c_void_p]
M, Lq = 14400, 450
st = torch.zeros(16384, dtype=torch.int64, device="cuda")
assert lib.tcdiff_debug_chain_stamp_buffer(st.data_ptr()) == 0
d = T._tail_inputs(M, Lq)
bf = torch.bfloat16
xo = torch.zeros(M, 512, device="cuda"); ho = torch.zeros(M, 512, device="cuda", dtype=bf); ro = torch.zeros(M, 512, device="cuda", dtype=bf)
fld = d["film"].shape[1]
def call():
    K.chain_tail(d["O"], d["Wfc"], d["lnp_g"], d["lnp_b"], d["film"][:, 1024:], d["film"][:, 2048:], fld, d["xres"], d["ln3_g"],
                 d["ln3_b"], d["W1"], d["b1"], d["W2"], d["b2"], d["ln4_g"], d["ln4_b"], d["W3"], d["b3"], xo, d["ln1_g"], d["ln1_b"], ho, ro, d["rope"], M, Lq)
for _ in range(3): call()
torch.cuda.synchronize()
s = st[:225 * 16].view(225, 16).cpu().double() * 10.0
names = ["gemm_fc(16 tiles)", "rowphase1", "ffn(96 tiles)", "rowphase2", "lin3(16 tiles)", "rowphase3"]
dd = s[:, 1:7] - s[:, 0:6]
print({n: round(float(v)) for n, v in zip(names, dd.mean(0))}, "total ns:", round(float((s[:, 6] - s[:, 0]).mean())))
raw = st[:225 * 16].view(225, 16).cpu().double()
clk = (raw[:, 9] - raw[:, 8]) / ((raw[:, 6] - raw[:, 0]) * 10.0)   # shader cycles per ns = GHz
print("shader clock during the kernel: median %.2f GHz (min %.2f, max %.2f)" % (float(clk.median()), float(clk.min()), float(clk.max())))
print("chain_tail device time (graph): %.1f us" % ev(call))

fine = st[8192:8192 + 2048].cpu().view(2, 128, 8)
for w in range(2):
    f = fine[w][:8].double()
    print("wave", w * 5, "per half-0 iteration (shader cycles): [wait vmcnt, barrier, issue dma+pf, reads+mfma] and gap to next iteration")
    for i in range(7):
        d = f[i, 1:5] - f[i, 0:4]
        print("   ", [int(x) for x in d], "next:", int(f[i + 1, 0] - f[i, 4]))
