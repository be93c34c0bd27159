"""Diagnostic: board power and shader clock (rocm-smi, polled every 0.2 s) while the chip runs ONE kind of work at a time --
nothing, the pure weight stream (tools/probe/l2_alias_probe: every CU streaming a layer's weights, no arithmetic), the
self-attention launch, the chain-B launch, the whole sampler -- to see which of them holds the board at its power limit.
python tools/power_parts.py [seconds per part]"""
import json, os, statistics, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tcdiff_amd import _lib as L, kernels as K
from tcdiff_amd.engine import DenoiserEngine as E

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
samples, stop = [], False


def poll():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            c = json.loads(out)
            c = c[sorted(c)[0]]
            pw = [float(v) for k, v in c.items() if "power" in k.lower() and "(w)" in k.lower()]
            sclk = [v for k, v in c.items() if k.lower().startswith("sclk clock speed")]
            samples.append((time.time(), pw[0] if pw else float("nan"), sclk[0] if sclk else "?"))
        except Exception as e:       # noqa: BLE001
            samples.append((time.time(), float("nan"), repr(e)[:60]))
        time.sleep(0.2)


dev, bf = "cuda", torch.bfloat16
torch.manual_seed(0)
Lq, H, Lp = 450, 8, 512
rnd = lambda *s, scale=1.0: torch.randn(*s, device=dev) * scale
W = {n: rnd(*s, scale=s[1] ** -0.5).to(bf) for n, s in [("cfc", (512, 512)), ("ff1", (1024, 512)), ("ff2", (512, 1024)), ("l3", (512, 512)),
                                                         ("qkv", (1536, 512))]}
vec = lambda base=0.0: base + 0.1 * rnd(512)
parts = [E._stages_n512(W["cfc"])] + E._ffn_order(E._stages_ff1(W["ff1"]), E._stages_ff2(W["ff2"])) + [E._stages_n512(W["l3"])] + \
    [E._stages_n512(W["qkv"][i * 512:(i + 1) * 512]) for i in range(3)]
wsB = torch.cat(parts, 1).contiguous()
rope = torch.empty(Lq, 512, device=dev)
K.rope_table((1.0 / (10000 ** (torch.arange(0, 512, 2).float() / 512))).to(dev), rope, Lq)
rope = K.to_cb(rope)
g = [vec(1), vec(), vec(1), vec(), vec(1), vec(), vec(1), vec()]
b1, b3 = 0.05 * rnd(1024), vec()
M = 225 * 64
nseq = 32
Oa, film, x = rnd(M, 512, scale=0.5).to(bf), 0.3 * rnd(nseq, 4096), rnd(M, 512)
Q, Kk, V = (torch.randn(nseq, H, Lp, 64, device=dev).to(bf) for _ in range(3))
O = torch.empty(nseq * Lq, 512, device=dev, dtype=bf)


def chain_b():
    K.chain(L.CHAIN_B, M, Lq, Oa, wsB, mt=4, ln_eps=1e-6, film=film, film_ld=4096, xres=x, xout=x, n2_g=g[2], n2_b=g[3], rope=rope,
            b1=b1, film3=film[:, 2048:], n4_g=g[4], n4_b=g[5], b3=b3, nn_g=g[6], nn_b=g[7], q_out=Q, k_out=Kk, v_out=V, Lp=Lp, H=H)


def attn():
    K.attention(L.DT_BF16, Q, Kk, V, O, nseq, H, Lq, Lq, Lp, Lp, 512)


def loop(fn, label):
    t0 = time.time()
    n = 0
    while time.time() - t0 < secs:
        for _ in range(200):
            fn()
        torch.cuda.synchronize()
        n += 200
    t1 = time.time()
    return label, t0, t1, f"{(t1 - t0) / n * 1e6:.1f} us per launch"


th = threading.Thread(target=poll, daemon=True)
th.start()
spans = []
t0 = time.time(); time.sleep(secs); spans.append(("idle (context up)", t0, time.time(), ""))
probe = os.path.join(ROOT, "tools", "probe", "l2_alias_probe")
for nb in (225, 256):
    t0 = time.time()
    out = subprocess.run([probe, "loop", str(secs), str(nb)], capture_output=True, text=True).stdout.strip()
    spans.append((f"pure weight stream, {nb} CUs", t0, time.time(), out))
spans.append(loop(attn, "self-attention launch (256 workgroups)"))
spans.append(loop(chain_b, "chain-B launch (225 blocks)"))
time.sleep(1.0)
stop = True
th.join()
for label, a, b, note in spans:
    pw = [p for t, p, _ in samples if a + 1.0 <= t <= b and p == p]
    ck = [c for t, p, c in samples if a + 1.0 <= t <= b]
    print(f"{label:42s} power median {statistics.median(pw) if pw else float('nan'):7.0f} W  (min {min(pw) if pw else 0:.0f}, max {max(pw) if pw else 0:.0f}, "
          f"{len(pw)} samples)  sclk {ck[len(ck) // 2] if ck else '?'}  {note}")
