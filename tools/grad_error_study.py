"""Diagnostic (CPU, test infrastructure: imports oracle/): where does the bf16 training step's gradient error come from
(VERDICT r3 #4d: worst parameter rel-L2 0.19 at the benchmarked config, self-attention w_qs / w_ks of the last layers)?
The oracle's forward is re-run with bf16 ROUNDING emulated at chosen points (values stay fp32 tensors, rounded to the bf16 grid):
  A  GEMM / attention OPERANDS rounded in the forward only (the backward differentiates that rounded forward in fp32):
     what ANY bf16-operand implementation pays, whatever its backward does
  B  A + every activation GRADIENT that the HIP step stores in bf16 rounded too (gradients of linear inputs / outputs,
     dQ / dK / dV / dO), fp32 parameter gradients and fp32 residual-stream gradient as in the engine
  C  B without A: exact forward, rounded gradient tensors only
  D  B + every GEMM OUTPUT rounded in the forward too (the engine stores z1 .. z4, Q / K / V and O in bf16)
and the per-parameter relative L2 distance to the fp32 gradients is printed.   python tools/grad_error_study.py [dn S B]"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as TF
from oracle import tcdiff_oracle as O

torch.set_num_threads(8)
dn, S, B = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 60, 2)
bf = lambda x: x.to(torch.bfloat16).to(torch.float32)


class RoundF(torch.autograd.Function):      # value rounded in the forward, gradient passed through
    @staticmethod
    def forward(ctx, x): return bf(x)
    @staticmethod
    def backward(ctx, g): return g


class RoundB(torch.autograd.Function):      # identity in the forward, gradient rounded
    @staticmethod
    def forward(ctx, x): return x.view_as(x)
    @staticmethod
    def backward(ctx, g): return bf(g)


MODE = {"fwd": False, "bwd": False, "out": False}
rf = lambda x: RoundF.apply(x) if MODE["fwd"] else x
rb = lambda x: RoundB.apply(x) if MODE["bwd"] else x


def linear(x, w, b=None):                   # operands bf16, fp32 accumulate; dX and dY live in bf16, dW in fp32
    y = rb(TF.linear(rf(rb(x)), rf(w), b))
    return RoundF.apply(y) if MODE["out"] else y      # D: the GEMM OUTPUT stored in bf16 as well (the engine's z1 .. z4, Q / K / V, O)


def matmul(a, b):                           # attention products: operands bf16; dQ / dK / dV / dP / dO in bf16
    return rb(torch.matmul(rf(rb(a)), rf(rb(b))))


Fp = types.SimpleNamespace(**{k: getattr(TF, k) for k in dir(TF) if not k.startswith("__")})
Fp.linear = linear
Tp = types.SimpleNamespace(**{k: getattr(torch, k) for k in dir(torch) if not k.startswith("__")})
Tp.matmul = matmul
O.F, O.torch = Fp, Tp                       # the oracle's own F.linear / torch.matmul calls go through the emulation

sd0 = O.synth_state_dict(dn=dn, seq_len=S)
x = torch.stack([O.synth_motion(c, dn * S) for c in range(B)])
cond = torch.stack([O.synth_cond(c, S) for c in range(B)])
noise = torch.stack([O.synth_xT(c, dn * S) for c in range(B)])
t = torch.tensor([(7 * c + 3) % 1000 for c in range(B)])
tab = O.make_tables(1000, "cosine")


def grads(fwd, bwd, out=False):
    MODE["fwd"], MODE["bwd"], MODE["out"] = fwd, bwd, out
    sd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd0.items()}
    xt = O.q_sample(tab, x, t, noise)
    out = O.decoder_forward(sd, xt, cond, t, keep_mask=torch.ones(B, dtype=torch.bool))
    loss = ((out - x) ** 2).mean()
    loss.backward()
    return float(loss), {k: v.grad.clone() for k, v in sd.items() if torch.is_tensor(v) and v.requires_grad and v.grad is not None}


l0, g0 = grads(False, False)
print(f"{dn} x {S}, batch {B}: fp32 loss {l0:.6f}, {len(g0)} parameters with gradients")
for name, (f_, b_, *rest) in (("A  forward operands rounded", (True, False)), ("B  A + gradient tensors rounded", (True, True)),
                       ("C  gradient tensors only", (False, True)), ("D  B + GEMM outputs stored in bf16", (True, True, True))):
    l1, g1 = grads(f_, b_, *rest)
    rel = {k: float((g1[k] - g0[k]).norm() / (g0[k].norm() + 1e-30)) for k in g0}
    v = np.array(sorted(rel.values()))
    worst = sorted(rel.items(), key=lambda kv: -kv[1])[:4]
    qk = [rel[k] for k in rel if k.endswith("self_attn.w_qs.weight") or k.endswith("self_attn.w_ks.weight")]
    print(f"{name:34s} loss {l1:.6f} | rel-L2: median {np.median(v):.3e}  90th {v[int(0.9 * len(v))]:.3e}  worst {v[-1]:.3e} | "
          f"self-attn w_qs / w_ks: median {np.median(qk):.3e} max {max(qk):.3e}")
    print("      worst: " + ", ".join(f"{k.replace('seqTransDecoder.stack.', 'L')} {e:.2e}" for k, e in worst))
