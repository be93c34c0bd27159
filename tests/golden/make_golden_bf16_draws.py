"""Draws of the bf16 training step's rounding noise, from the CPU oracle alone (no reference import: the oracle is pinned to the
reference's gradients by tests/golden/c1_train_step.npz; this file only characterises how far a bf16 step lies from the exact one).

The bf16 step's distance to the exact fp32 gradient is the forward's operand rounding amplified by the loss: a
chaotic quantity -- ONE emulated evaluation is one draw, and round 5's GPU test compared the kernels with one draw (kernels / draw =
1.76 at 32 clips, unexplained).  This script evaluates the oracle with the kernels' rounding points (oracle.operand_rounding) several
times, each draw perturbed immaterially (x_start scaled by 1 + k 2^-18: the exact gradient moves by ~4e-6 k relative, the rounding
pattern completely), and stores per draw and per parameter the relative L2 distance to the exact gradient:

    python tests/golden/make_golden_bf16_draws.py [--b 3 32] [--draws 6]   ->   tests/golden/c5_bf16_draws.npz

tests/test_train_step_gpu.py::test_bf16_step_error_profile_is_a_draw_of_the_emulated_rounding_noise holds the HIP step to the
distribution: per-parameter SHAPE of the error (parameter error / median error) and its SCALE."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import tcdiff_oracle as O   # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def inputs(b, dn=3, S_=150):
    x_start = torch.stack([O.synth_motion(300 + c, dn * S_).reshape(S_, dn, 151).permute(1, 0, 2) for c in range(b)])
    cond = torch.stack([O.synth_cond(300 + c, S_) for c in range(b)])
    noise = torch.stack([O.synth_xT(300 + c, dn * S_).reshape(S_, dn, 151) for c in range(b)])
    g = torch.Generator().manual_seed(77)
    t = torch.randint(0, 1000, (b,), generator=g)
    keep = torch.rand(b, generator=g) > 0.25
    return x_start, cond, noise, t, keep


def grads(sd, b, mode, k=0, seed=(2024, 1003)):
    x_start, cond, noise, t, keep = inputs(b)
    x_start = x_start * (1.0 + k * 2.0 ** -18)
    sd_now = {n: p.detach().clone().requires_grad_(True) for n, p in sd.items() if p.is_floating_point()}
    ctx = O.operand_rounding(**mode) if mode is not None else torch.enable_grad()
    with ctx:
        total, _ = O.p_losses(sd_now, O.make_tables(1000), x_start, cond, t, noise, keep, drop=O.DropPlan(seed, 0.1))
        total.backward()
    return {n: v.grad.numpy().copy() for n, v in sd_now.items() if v.grad is not None}


def c1_grads(sd, ref, k, mode, d=0):
    """step k of tests/golden/c1_train_step.npz's two steps (config-1 shape, 3 clips of 2 x 60; step 0 eval, step 1 train mode with
    the weights after Adan's first step, which only decays them: model/adan.py:71,96-107) -- test_train_step_gpu.step_inputs"""
    DN, S, B = 2, 60, 3
    noise0 = (10, 20)[k]
    x_start = torch.stack([O.synth_motion(100 * k + c, DN * S).reshape(S, DN, 151).permute(1, 0, 2) for c in range(B)])
    cond = torch.stack([O.synth_cond(100 * k + c, S) for c in range(B)])
    noise = torch.stack([O.synth_xT(noise0 + c, DN * S).reshape(S, DN, 151) for c in range(B)])
    t, keep = torch.from_numpy(ref[f"s{k}_t"]), torch.from_numpy(ref[f"s{k}_keep"])
    seed = tuple(int(v) for v in ref[f"s{k}_seed"])
    train = bool(ref[f"s{k}_train"])
    x_start = x_start * (1.0 + d * 2.0 ** -18)
    decay = 1.0 / (1.0 + float(ref["lr"]) * float(ref["wd"])) if k == 1 else 1.0
    sd_now = {n: (p.detach().clone() * decay).requires_grad_(True) for n, p in sd.items() if p.is_floating_point()}
    ctx = O.operand_rounding(**mode) if mode is not None else torch.enable_grad()
    with ctx:
        total, _ = O.p_losses(sd_now, O.make_tables(100), x_start, cond, t, noise, keep,
                              drop=O.DropPlan(seed, float(ref["p_drop"]) if train else 0.0))
        total.backward()
    return {n: v.grad.numpy().copy() for n, v in sd_now.items() if v.grad is not None}


def norot_grads(sd, mode, d=0):
    """the train-mode step of test_training_step_without_rotary_embedding_vs_oracle_autograd (config-1 shape, use_rotary=False)"""
    DN, S, B = 2, 60, 3
    x_start = torch.stack([O.synth_motion(c, DN * S).reshape(S, DN, 151).permute(1, 0, 2) for c in range(B)]) * (1.0 + d * 2.0 ** -18)
    cond = torch.stack([O.synth_cond(c, S) for c in range(B)])
    noise = torch.stack([O.synth_xT(10 + c, DN * S).reshape(S, DN, 151) for c in range(B)])
    t, keep, seed = torch.tensor([17, 80, 3]), torch.tensor([True, False, True]), (11, 5)
    sd_now = {n: p.detach().clone().requires_grad_(True) if p.is_floating_point() and n != "abs_pos_encoding.pe" else p
              for n, p in sd.items()}
    ctx = O.operand_rounding(**mode) if mode is not None else torch.enable_grad()
    with ctx:
        total, _ = O.p_losses(sd_now, O.make_tables(100), x_start, cond, t, noise, keep, drop=O.DropPlan(seed, 0.1))
        total.backward()
    return {n: v.grad.detach().numpy().copy() for n, v in sd_now.items() if v.requires_grad and v.grad is not None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--c1", action="store_true", help="also the two steps of c1_train_step.npz (3 clips of 2 x 60)")
    ap.add_argument("--norot", action="store_true", help="also the use_rotary=False step (3 clips of 2 x 60, train mode)")
    ap.add_argument("--b", type=int, nargs="*", default=[3, 32])
    ap.add_argument("--draws", type=int, default=6)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "c5_bf16_draws.npz"))
    a = ap.parse_args()
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    sd = O.synth_state_dict(dn=3, seq_len=150)
    MODES = {"A": dict(fwd=True, bwd=False), "D": dict(fwd=True, bwd=True, out=True)}
    out = {}
    if os.path.exists(a.out):
        out = dict(np.load(a.out, allow_pickle=False))
    for b in a.b:
        t0 = time.time()
        g0 = grads(sd, b, None)
        names = sorted(g0)
        print(f"b={b}: exact gradients of {len(names)} parameters in {time.time() - t0:.0f} s", flush=True)
        E, desc = [], []
        for d in range(a.draws):
            m = "D" if d % 3 else "A"          # draws 0, 3: forward operands only; the others: every rounding point of the kernels
            t1 = time.time()
            ge = grads(sd, b, MODES[m], k=d)
            e = np.array([rel(ge[n], g0[n]) for n in names], np.float32)
            E.append(e)
            desc.append(f"{m}{d}")
            print(f"b={b} draw {desc[-1]}: median {np.median(e):.3e} worst {e.max():.3e} ({names[int(e.argmax())]}) in {time.time() - t1:.0f} s",
                  flush=True)
        out[f"b{b}_names"] = np.array(names)
        out[f"b{b}_err"] = np.stack(E)
        out[f"b{b}_draws"] = np.array(desc)
        np.savez_compressed(a.out, **out)
    if a.c1:
        ref = np.load(os.path.join(ROOT, "tests", "golden", "c1_train_step.npz"))
        sd1 = O.synth_state_dict(dn=2, seq_len=60)
        for k in (0, 1):
            g0 = c1_grads(sd1, ref, k, None)
            names = sorted(g0)
            E, desc = [], []
            for d in range(a.draws):
                m = "D" if d % 3 else "A"
                ge = c1_grads(sd1, ref, k, MODES[m], d)
                e = np.array([rel(ge[n], g0[n]) for n in names], np.float32)
                E.append(e)
                desc.append(f"{m}{d}")
                print(f"c1 step {k} draw {desc[-1]}: median {np.median(e):.3e} worst {e.max():.3e} ({names[int(e.argmax())]})", flush=True)
            out[f"c1s{k}_names"], out[f"c1s{k}_err"], out[f"c1s{k}_draws"] = np.array(names), np.stack(E), np.array(desc)
        np.savez_compressed(a.out, **out)
    if a.norot:
        sdn = O.synth_state_dict(dn=2, seq_len=60, use_rotary=False)
        g0 = norot_grads(sdn, None)
        names = sorted(g0)
        E, desc = [], []
        for d in range(a.draws):
            m = "D" if d % 3 else "A"
            ge = norot_grads(sdn, MODES[m], d)
            e = np.array([rel(ge[n], g0[n]) for n in names], np.float32)
            E.append(e)
            desc.append(f"{m}{d}")
            print(f"use_rotary=False draw {desc[-1]}: median {np.median(e):.3e} worst {e.max():.3e} ({names[int(e.argmax())]})", flush=True)
        out["norot_names"], out["norot_err"], out["norot_draws"] = np.array(names), np.stack(E), np.array(desc)
        np.savez_compressed(a.out, **out)
    print("saved", a.out)


if __name__ == "__main__":
    main()
