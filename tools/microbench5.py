"""diagnostic (TC_STAMP build): phases of one attention workgroup"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tcdiff_amd import _lib as L, kernels as K
from tools.microbench import ev
lib = L.load(); lib.tcdiff_debug_attn_stamp_buffer.argtypes = [ctypes.c_void_p]
st = torch.zeros(4 * 128, dtype=torch.int64, device="cuda"); assert lib.tcdiff_debug_attn_stamp_buffer(st.data_ptr()) == 0
dev, dt, T = "cuda", L.DT_BF16, torch.bfloat16
nseq, H, Lq, Lp = 32, 8, 450, 512
Q = torch.randn(nseq, H, Lp, 64, device=dev).to(T) * 0.3; Kk = torch.randn(nseq, H, Lp, 64, device=dev).to(T); V = torch.randn(nseq, H, Lp, 64, device=dev).to(T)
O = torch.zeros(nseq * Lq, 512, device=dev, dtype=T)
for _ in range(3): K.attention(dt, Q, Kk, V, O, nseq, H, Lq, Lq, Lp, Lp, 512)
torch.cuda.synchronize()
f = st.cpu().view(4, 16, 8).double()
for w in (0, 3):
    print("wave", w, "per KV iteration (shader cycles): [issue loads, S=KQ^T, softmax, PV, wait+store, barrier], gap")
    for b in range(7):
        d = f[w, b, 1:7] - f[w, b, 0:6]
        print("   ", [int(x) for x in d], "total", int(f[w, b + 1, 0] - f[w, b, 0]))
print("attention device time: %.1f us" % ev(lambda: K.attention(dt, Q, Kk, V, O, nseq, H, Lq, Lq, Lp, Lp, 512)))
