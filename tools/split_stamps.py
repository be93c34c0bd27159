"""Diagnostic: time line of the four launches of the small-job layer (block 0, one member) from a -DCS_STAMP build
(tools/ab_build_split.sh STAMPS "-DCS_STAMP"; TCDIFF_LIB_PATH=tools/probe/libtc_STAMPS.so python tools/split_stamps.py)."""
import os
import sys

os.environ.setdefault("TCDIFF_LIB_PATH", "tools/probe/libtc_STAMPS.so")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch  # noqa: E402
from test_chain_split_gpu import run_layers  # noqa: E402

NAMES = {
    1: ["entry", "attention done", "barrier", "fc quarter", "partial stored"],
    2: ["entry", "loads issued", "epilogue (sum, LN, FiLM, +x)", "norm2 + rot -> LDS, barrier", "w_qs head projection", "cross-attention", "barrier",
        "fc quarter", "partial stored"],
    3: ["entry", "loads issued", "epilogue", "norm3 -> LDS, barrier", "linear1 chunk", "GELU -> LDS, barrier", "linear2 chunk", "partial stored"],
    4: ["entry", "loads issued", "epilogue", "norm4 -> LDS, barrier", "linear3", "x' stored", "norm1' + rot -> LDS, barrier", "Q", "K", "V"],
}
Lq, nseq, Lk = int(os.environ.get("LQ", 150)), int(os.environ.get("NSEQ", 3)), int(os.environ.get("LK", 152))
st = torch.zeros(4 * 128, dtype=torch.int64, device="cuda")
for _ in range(3):
    run_layers(Lq, nseq, True, 1.0, Lk, stamps=st)
t = st.cpu().view(4, 8, 16)
print(f"small-job layer, {nseq} x {Lq} rows, {Lk} keys: block 0, wave 0's stamps (us since entry; 100 MHz counter), slowest wave in brackets")
for part in (1, 2, 3, 4):
    base = int(t[part - 1, :, 0].min())
    names = NAMES[part]
    print(f"  part {part}:")
    prev = 0.0
    for i, n in enumerate(names):
        w0 = (int(t[part - 1, 0, i]) - base) / 100.0
        worst = (int(t[part - 1, :, i].max()) - base) / 100.0
        print(f"    {n:44s} {w0:7.2f}  (+{w0 - prev:5.2f})   [{worst:7.2f}]")
        prev = w0
