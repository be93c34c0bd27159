"""The training step on MI355X (through the C ABI) against the REAL reference's gradients.

tests/golden/c1_train_step.npz holds two consecutive steps of TCDiff.train_loop's body (TCDiff.py:227-234) run on the real
model/model.py + model/diffusion.py + model/adan.py with every random draw injected (tests/golden/make_golden_train_step.py):
step 0 in eval mode, step 1 in train mode with the dropout masks of the product's counter hash.  Here the same two steps run
on the HIP path -- `total, _ = diffusion.p_losses(...)`, `optim.zero_grad()`, `total.backward()`, `optim.step()` -- and are
compared with (a) the golden's sampled gradients / parameters of 37 named parameters and (b) the CPU oracle's full gradients
of EVERY parameter (the oracle's autograd agrees with the reference's to <= 2e-5 relative L2 on these inputs: the generator's log)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import tcdiff_oracle as O  # noqa: E402  (checker only)
from tcdiff_amd import Adan  # noqa: E402
from tcdiff_amd.diffusion import GaussianDiffusion  # noqa: E402
from tcdiff_amd.model import DanceDecoder  # noqa: E402

DEV = "cuda"
DN, S, T, B = 2, 60, 100, 3


def build(compute, sd=None, dn=DN, S_=S, T_=T):
    sd = O.synth_state_dict(dn=dn, seq_len=S_) if sd is None else sd
    model = DanceDecoder(nfeats=151, seq_len=S_, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                         cond_feature_dim=438, activation=F.gelu, required_dancer_num=dn, compute_dtype=compute)
    model.load_state_dict(sd)
    diff = GaussianDiffusion(model, S_, 151, None, schedule="cosine", n_timestep=T_, predict_epsilon=False, loss_type="l2",
                             use_p2=False, cond_drop_prob=0.25, guidance_weight=2, seq_len=S_)
    return sd, diff.to(DEV)


def step_inputs(k, noise0):
    x_start = torch.stack([O.synth_motion(100 * k + c, DN * S).reshape(S, DN, 151).permute(1, 0, 2) for c in range(B)])
    cond = torch.stack([O.synth_cond(100 * k + c, S) for c in range(B)])
    noise = torch.stack([O.synth_xT(noise0 + c, DN * S).reshape(S, DN, 151) for c in range(B)])
    return x_start, cond, noise


def sample(t: torch.Tensor) -> np.ndarray:
    f = t.detach().reshape(-1).cpu()
    return f[::max(1, f.numel() // 4096)].numpy()


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


# gradient tolerance (relative L2 per parameter): f32 mode is the parity mode (the north-star's fp32 claim); the bf16 mode
# rounds every GEMM / attention operand and every T-typed activation gradient to 8 bits of mantissa
# bf16 observed (worst parameter): 1.1e-1 .. 1.5e-1 with the decoder linears through gemm_tile, 1.5e-1 .. 2.1e-1 through
# gemm_rows (the default) -- two draws of the same forward rounding noise, see
# test_row_block_gemm_path_and_tile_path_are_two_roundings_of_the_same_step
TOL = {"f32": (1e-4, 2e-5), "bf16": (2.5e-1, 2e-2)}      # (per-parameter gradient, loss terms)
# Round 6: the bf16 gradient bounds are DERIVED, not chosen: tests/golden/c5_bf16_draws.npz holds nine draws of the oracle evaluated with
# the kernels' rounding points per configuration (make_golden_bf16_draws.py); a bf16 step's worst parameter may lie 1.35 x above the
# worst draw's worst parameter (nine draws do not exhaust a distribution), and every parameter's error RELATIVE TO THE MEDIAN must follow
# the draws' common profile (SHAPE_TOL; test_bf16_step_error_profile_is_a_draw_of_the_emulated_rounding_noise).  TOL["bf16"][0] remains
# as the ceiling no derived bound may exceed.
SHAPE_TOL = 1.3


def bf16_bound(golden_dir, key):
    """(worst-parameter bound, names, per-draw error matrix) of configuration `key` in c5_bf16_draws.npz"""
    z = np.load(os.path.join(golden_dir, "c5_bf16_draws.npz"))
    E = z[f"{key}_err"].astype(np.float64)
    return min(TOL["bf16"][0] if key.startswith("c1") else 4e-1, 1.35 * float(E.max())), [str(n) for n in z[f"{key}_names"]], E


def shape_ratio(err_by_name, names, E):
    """max over the parameters of (error / median error) / (the draws' envelope of the same quantity)"""
    idx = [i for i, n in enumerate(names) if n in err_by_name and E[:, i].max() > 0]
    ek = np.array([err_by_name[names[i]] for i in idx])
    Ed = E[:, idx]
    hi = (Ed / np.median(Ed, axis=1)[:, None]).max(axis=0)
    r = (ek / np.median(ek)) / hi
    j = int(r.argmax())
    return float(r[j]), names[idx[j]], float(np.median(ek)), np.median(Ed, axis=1)


@pytest.mark.parametrize("compute", ["f32", "bf16"])
def test_two_training_steps_vs_reference_golden_and_oracle(golden_dir, compute):
    ref = np.load(os.path.join(golden_dir, "c1_train_step.npz"))
    names = [str(n) for n in ref["names"]]
    gtol, ltol = TOL[compute]
    sd, diff = build(compute)
    model = diff.model
    optim = Adan(model.parameters(), lr=float(ref["lr"]), weight_decay=float(ref["wd"]))
    named = dict(model.named_parameters())
    tab = O.make_tables(T)
    worst = {}
    for k, noise0 in ((0, 10), (1, 20)):
        train = bool(ref[f"s{k}_train"])
        diff.train(train)
        x_start, cond, noise = step_inputs(k, noise0)
        t, keep = torch.from_numpy(ref[f"s{k}_t"]), torch.from_numpy(ref[f"s{k}_keep"])
        seed = tuple(int(v) for v in ref[f"s{k}_seed"])
        model.train_seed = seed
        total, losses = diff.p_losses(x_start.to(DEV), cond.to(DEV), t.to(DEV), noise=noise.to(DEV), keep_mask=keep.to(DEV))
        assert total.requires_grad and total.grad_fn is not None
        optim.zero_grad()
        total.backward()
        got_l = np.array([float(v) for v in losses])
        print(f"[{compute}] step {k} ({'train' if train else 'eval'}): total {float(total):.6f} (reference "
              f"{float(ref[f's{k}_total']):.6f}); losses {np.round(got_l, 6)} (reference {np.round(ref[f's{k}_losses'], 6)})")
        assert np.all(np.abs(got_l - ref[f"s{k}_losses"]) <= ltol * np.maximum(np.abs(ref[f"s{k}_losses"]), 1e-3))
        # (a) the real reference's gradients (sampled) and gradient norms
        for n in names:
            g = named[n].grad
            assert g is not None, n
            r1 = rel(sample(g), ref[f"s{k}_g:{n}"])
            rn = abs(float(g.norm()) - float(ref[f"s{k}_gn:{n}"])) / float(ref[f"s{k}_gn:{n}"])
            worst[(k, n)] = r1
        top = sorted(((v, n) for (kk, n), v in worst.items() if kk == k), reverse=True)
        print(f"[{compute}] step {k}: sampled gradients vs the REFERENCE, rel-L2: worst " +
              ", ".join(f"{n} {v:.2e}" for v, n in top[:4]) + f"; median {np.median([v for v, _ in top]):.2e}")
        for n in names:
            g = named[n].grad
            rn = abs(float(g.norm()) - float(ref[f"s{k}_gn:{n}"])) / float(ref[f"s{k}_gn:{n}"])
            assert worst[(k, n)] < gtol and rn < gtol, (k, n, worst[(k, n)], rn)
        # (b) every parameter against the oracle's autograd on the same weights and draws (for step 1 the weights are
        # the HIP path's own after its first Adan step)
        sd_now = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in model.state_dict().items() if p.is_floating_point()}
        plan = O.DropPlan(seed, float(ref["p_drop"]) if train else 0.0)
        o_total, _ = O.p_losses(sd_now, tab, x_start, cond, t, noise, keep, drop=plan)
        o_total.backward()
        n_dead, allr = 0, []
        for n, p in named.items():
            og = sd_now[n].grad
            if p.grad is None:
                n_dead += 1
                assert og is None or float(og.abs().max()) == 0.0, n         # unused by the forward in the reference too
                continue
            allr.append((rel(p.grad.cpu().numpy(), og.numpy()), n))
        allr.sort(reverse=True)
        print(f"[{compute}] step {k}: ALL {len(allr)} parameter gradients vs the oracle's autograd, rel-L2: worst " +
              ", ".join(f"{n} {v:.2e}" for v, n in allr[:4]) + f"; median {np.median([v for v, _ in allr]):.2e}")
        if compute == "bf16":
            bound, dn_names, E = bf16_bound(golden_dir, f"c1s{k}")
            sr, sn, med_k, med_d = shape_ratio({n: v for v, n in allr}, dn_names, E)
            print(f"[bf16] step {k}: derived bound {bound:.2e} (1.35 x the nine draws' worst {E.max():.2e}); error shape against the draws' "
                  f"envelope: max {sr:.2f} ({sn}); median {med_k:.2e} vs the draws' {med_d.min():.2e} .. {med_d.max():.2e}")
            # What the pin found (round 6) and what became of it: at THIS shape, eval mode (the lowest noise floor of the file), the
            # decoder's self-attention w_qs / w_ks gradients of layers 4-7 lay 1.4e-1 .. 1.5e-1 from the exact ones where the nine draws
            # stay below 8.4e-2 -- 2.1 x the draws' profile, every other parameter within 1.3 x.  Owner: the backward's row term
            # delta_i = sum_d dO_id O_id taken from the 8-bit O image; it enters dS = P (dP - delta), a cancellation.  The oracle with
            # exactly that (oracle._AttnKernelDelta) reproduces the numbers, parameter by parameter.  Fix: the attention forward keeps
            # what O's 8 bits dropped (tcdiff_attention_train's O_lo), delta reads O + O_lo: worst parameter 8.4e-2, shape 1.02.
            assert allr[0][0] < bound and sr <= SHAPE_TOL and 0.5 * med_d.min() <= med_k <= 1.25 * med_d.max(), (allr[0], sr, sn, med_k)
        assert allr[0][0] < gtol, allr[0]
        assert n_dead == int(ref[f"s{k}_n_dead"]) == 125
        optim.step()
        # the first Adan step only decays the weights (model/adan.py:71): exact.  The second moves every element by lr times a
        # RATIO of gradient moments, (m^ + (1 - b2) v^) / (sqrt(n^) + eps), which is unbounded where n^ ~ 0: a few elements
        # amplify the relative error of their (tiny) gradient by orders of magnitude.  So the bound is on the mean and on a
        # high percentile, in units of lr = 5e-5.
        d = np.concatenate([np.abs(sample(named[n]) - ref[f"s{k}_p:{n}"]) for n in names])
        lr = float(ref["lr"])
        print(f"[{compute}] step {k}: parameters after Adan.step vs the reference: mean-abs {d.mean():.2e}, 99.9th percentile "
              f"{np.percentile(d, 99.9):.2e}, max {d.max():.2e} (lr = {lr:.0e})")
        if compute == "f32":
            assert d.mean() <= 0.002 * lr and np.percentile(d, 99.9) <= 0.05 * lr and d.max() <= 5 * lr
        else:
            assert d.mean() <= 0.3 * lr and np.percentile(d, 99.9) <= 20 * lr


@pytest.mark.parametrize("compute,b", [("f32", 32), ("bf16", 32), ("f32", 3)])
def test_training_step_at_the_benchmarked_config_vs_oracle_autograd(golden_dir, compute, b):
    """BASELINE config 5's per-GPU work -- 32 clips of 3 dancers x 150 frames, dropout 0.1 -- is the shape at which the step
    takes its fast paths: 14 400 token rows (a multiple of the k-tile: weight gradients by tcdiff_gemm_tn straight from the
    token-major operands), 450-token sequences (K/V-resident attention forward, operand-resident attention backward), fused
    activation epilogues.  One train-mode step against the oracle's autograd on the same weights, draws and dropout masks:
    every live parameter's gradient, f32 <= 1e-4 relative L2 (the parity mode), bf16 within its stated bound."""
    dn, S_ = 3, 150          # b = 3: 1350 token rows, not a multiple of the k-tile -> the repack + split-K weight-gradient path
    gtol, ltol = TOL[compute]
    sd, diff = build(compute, dn=dn, S_=S_, T_=1000)
    model = diff.model
    diff.train()
    x_start = torch.stack([O.synth_motion(300 + c, dn * S_).reshape(S_, dn, 151).permute(1, 0, 2) for c in range(b)])
    cond = torch.stack([O.synth_cond(300 + c, S_) for c in range(b)])
    noise = torch.stack([O.synth_xT(300 + c, dn * S_).reshape(S_, dn, 151) for c in range(b)])
    g = torch.Generator().manual_seed(77)
    t = torch.randint(0, 1000, (b,), generator=g)
    keep = torch.rand(b, generator=g) > 0.25
    seed = (2024, 1003)
    model.train_seed = seed
    total, losses = diff.p_losses(x_start.to(DEV), cond.to(DEV), t.to(DEV), noise=noise.to(DEV), keep_mask=keep.to(DEV))
    for p in model.parameters():
        p.grad = None
    total.backward()
    eng = model.train_engine()
    assert K_tn_taken(eng, b * dn * S_) == (b == 32)
    named = dict(model.named_parameters())
    if ("c5", b) not in _ORACLE_CACHE:         # the same weights and draws in both modes: one oracle evaluation (~1.5 minutes of CPU)
        sd_now = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in sd.items() if p.is_floating_point()}
        torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
        o_total, o_losses = O.p_losses(sd_now, O.make_tables(1000), x_start, cond, t, noise, keep, drop=O.DropPlan(seed, 0.1))
        o_total.backward()
        _ORACLE_CACHE[("c5", b)] = (float(o_total), np.array([float(v) for v in o_losses]),
                               {n: (None if v.grad is None else v.grad.numpy().copy()) for n, v in sd_now.items()})
    o_total, want_l, ograd = _ORACLE_CACHE[("c5", b)]
    got_l = np.array([float(v) for v in losses])
    assert np.all(np.abs(got_l - want_l) <= ltol * np.maximum(np.abs(want_l), 1e-3)), (got_l, want_l)
    allr = []
    for n, p in named.items():
        og = ograd[n]
        if p.grad is None:
            assert og is None or float(np.abs(og).max()) == 0.0, n
            continue
        allr.append((rel(p.grad.cpu().numpy(), og), n))
    allr.sort(reverse=True)
    print(f"[{compute}] config-5 shape ({b} x 3 x 150): total {float(total):.6f} (oracle {float(o_total):.6f}); ALL {len(allr)} parameter "
          f"gradients vs the oracle's autograd, rel-L2: worst " + ", ".join(f"{n} {v:.2e}" for v, n in allr[:4]) +
          f"; median {np.median([v for v, _ in allr]):.2e}")
    # 32 x 450 token rows: every weight gradient is an fp32 sum over 14 400 rows on BOTH sides (split-K partial sums here, torch's
    # blocked CPU summation in the oracle), so the f32 bound is 3x the small-config one; observed worst 1.2e-4 (layer 7's
    # cross-attention w_ks / w_qs, the smallest gradients of the model), median 2e-5
    # bf16: observed worst 1.9e-1 (layer 7 self-attention w_qs / w_ks), median 5e-2.  That distance is the FORWARD's bf16 operand
    # rounding, not the backward kernels': the CPU oracle with rounding emulated in the forward only (exact fp32 backward) lands
    # the same parameters at 1.45e-1 .. 1.94e-1 (tools/grad_error_study.py, profiles/r04_grad_error_study.txt)
    # (through gemm_rows, the default since round 4: worst 3.0e-1, median 8e-2 -- another draw of the same noise)
    bound = 3e-4 if compute == "f32" else bf16_bound(golden_dir, f"b{b}")[0]      # bf16: 1.35 x the worst of the nine emulated draws, capped at 4e-1
    assert len(allr) == 435 - 125 and allr[0][0] < bound, (allr[0], bound)


@pytest.mark.parametrize("name", ["relu", "silu"])
def test_training_step_with_another_feed_forward_activation_vs_oracle_autograd(name):
    """DanceDecoder(activation=F.relu | F.silu): one train-mode step (dropout live) at the C1 shape in the f32 mode, every live
    parameter's gradient against the oracle's autograd with the same activation (oracle.FF_ACTIVATION): <= 1e-4 relative L2."""
    fn = {"relu": F.relu, "silu": F.silu}[name]
    sd = O.synth_state_dict(dn=DN, seq_len=S)
    model = DanceDecoder(nfeats=151, seq_len=S, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                         cond_feature_dim=438, activation=fn, required_dancer_num=DN, compute_dtype="f32")
    model.load_state_dict(sd)
    diff = GaussianDiffusion(model, S, 151, None, schedule="cosine", n_timestep=T, predict_epsilon=False, loss_type="l2",
                             use_p2=False, cond_drop_prob=0.25, guidance_weight=2, seq_len=S).to(DEV)
    diff.train()
    x_start, cond, noise = step_inputs(0, 10)
    t, keep, seed = torch.tensor([17, 80, 3]), torch.tensor([True, False, True]), (11, 5)
    model.train_seed = seed
    total, _ = diff.p_losses(x_start.to(DEV), cond.to(DEV), t.to(DEV), noise=noise.to(DEV), keep_mask=keep.to(DEV))
    total.backward()
    sd_now = {n: p.detach().clone().requires_grad_(True) for n, p in sd.items() if p.is_floating_point()}
    O.FF_ACTIVATION = fn
    try:
        o_total, _ = O.p_losses(sd_now, O.make_tables(T), x_start, cond, t, noise, keep, drop=O.DropPlan(seed, 0.1))
        o_total.backward()
    finally:
        O.FF_ACTIVATION = None
    allr = sorted(((rel(p.grad.cpu().numpy(), sd_now[n].grad.numpy()), n) for n, p in model.named_parameters() if p.grad is not None),
                  reverse=True)
    print(f"[f32, activation={name}] total {float(total):.6f} (oracle {float(o_total):.6f}); {len(allr)} gradients vs the oracle's autograd: "
          f"worst {allr[0][1]} {allr[0][0]:.2e}, median {np.median([v for v, _ in allr]):.2e}")
    assert abs(float(total) - float(o_total)) <= 2e-5 * abs(float(o_total)) and allr[0][0] < 1e-4


@pytest.mark.parametrize("compute,train", [("f32", True), ("f32", False), ("bf16", True)])
def test_training_step_without_rotary_embedding_vs_oracle_autograd(golden_dir, compute, train):
    """DanceDecoder(use_rotary=False) (model/model.py:441-448): no rotation anywhere and PositionalEncoding -- with its own nn.Dropout,
    model/utils.py:27-32 -- on the motion tokens (:564) and the music tokens (:580).  One step at the C1 shape, dropout live (the two
    extra sites 8 / 9 of the DropPlan: same counter-hash masks on both sides) and in eval mode, every live parameter's gradient
    against the oracle's autograd; the oracle's use_rotary=False forward is pinned to the real reference by c1_abs_pos.npz
    (test_parity_gpu.py).  f32 <= 1e-4 relative L2 -- in eval mode the self-attention of the last layers saturates with these weights
    and its w_qs / w_ks gradients vanish (|g| 3e-8 against a median of 7e-3): those are held to an absolute floor of 1e-10 per element,
    fp32 summation noise.  bf16: this configuration amplifies rounding noise more than the rotary one (nine emulated draws,
    tests/golden/c5_bf16_draws.npz `norot`: median error 0.11 .. 0.76) -- the step is held to the draws' per-parameter profile and range."""
    sd = O.synth_state_dict(dn=DN, seq_len=S, use_rotary=False)
    model = DanceDecoder(nfeats=151, seq_len=S, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                         cond_feature_dim=438, activation=F.gelu, required_dancer_num=DN, use_rotary=False, compute_dtype=compute)
    model.load_state_dict(sd)
    diff = GaussianDiffusion(model, S, 151, None, schedule="cosine", n_timestep=T, predict_epsilon=False, loss_type="l2",
                             use_p2=False, cond_drop_prob=0.25, guidance_weight=2, seq_len=S).to(DEV)
    diff.train(train)
    x_start, cond, noise = step_inputs(0, 10)
    t, keep, seed = torch.tensor([17, 80, 3]), torch.tensor([True, False, True]), (11, 5)
    model.train_seed = seed
    total, _ = diff.p_losses(x_start.to(DEV), cond.to(DEV), t.to(DEV), noise=noise.to(DEV), keep_mask=keep.to(DEV))
    total.backward()
    assert model.train_engine().abs_pos
    sd_now = {n: p.detach().clone().requires_grad_(True) if p.is_floating_point() and n != "abs_pos_encoding.pe" else p
              for n, p in sd.items()}
    o_total, _ = O.p_losses(sd_now, O.make_tables(T), x_start, cond, t, noise, keep, drop=O.DropPlan(seed, 0.1 if train else 0.0))
    o_total.backward()
    def rel_floor(a, b):          # relative L2 with an absolute floor of 1e-10 per element under the reference's norm
        a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
        return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-10 * np.sqrt(b.size) / TOL["f32"][0]))
    allr = sorted(((rel_floor(p.grad.cpu().numpy(), sd_now[n].grad.numpy()), n) for n, p in model.named_parameters() if p.grad is not None),
                  reverse=True)
    gn = {n: float(sd_now[n].grad.norm()) for _, n in allr}
    print(f"[{compute}, use_rotary=False, {'train' if train else 'eval'}] total {float(total):.6f} (oracle {float(o_total):.6f}); "
          f"{len(allr)} gradients vs the oracle's autograd: worst " + ", ".join(f"{n} {v:.2e} (|g| {gn[n]:.1e})" for v, n in allr[:6]) +
          f"; median {np.median([v for v, _ in allr]):.2e}, median |g| {np.median(list(gn.values())):.1e}")
    gtol, ltol = TOL[compute]
    assert abs(float(total) - float(o_total)) <= ltol * abs(float(o_total)) and len(allr) > 300
    if compute == "f32":
        assert allr[0][0] < gtol, allr[0]
        return
    z = np.load(os.path.join(golden_dir, "c5_bf16_draws.npz"))
    names, E = [str(n) for n in z["norot_names"]], z["norot_err"].astype(np.float64)
    sr, sn, med_k, med_d = shape_ratio({n: v for v, n in allr}, names, E)
    print(f"[bf16, use_rotary=False] error profile vs nine emulated draws: shape {sr:.2f} ({sn}), median {med_k:.2e} (draws {med_d.min():.2e} .. "
          f"{med_d.max():.2e}), worst {allr[0][0]:.2e} (draws' worst {E.max():.2e})")
    assert sr <= SHAPE_TOL and 0.5 * med_d.min() <= med_k <= 1.25 * med_d.max() and allr[0][0] < 1.35 * float(E.max()), (sr, sn, med_k, allr[0])


def test_row_block_gemm_path_and_tile_path_are_two_roundings_of_the_same_step():
    """The decoder layers' linears run through tcdiff_gemm_rows (default) or tcdiff_gemm_tile (TCDIFF_TRAIN_ROWS=0): the same
    bf16 operands and fp32 accumulation, another summation order.  One train-mode step (3 clips of 3 x 150) both ways against
    the oracle's autograd: each path's distance to the oracle, and the two paths' distance to each other."""
    from tcdiff_amd import train_engine as TE
    dn, S_, b = 3, 150, 3
    x_start = torch.stack([O.synth_motion(300 + c, dn * S_).reshape(S_, dn, 151).permute(1, 0, 2) for c in range(b)])
    cond = torch.stack([O.synth_cond(300 + c, S_) for c in range(b)])
    noise = torch.stack([O.synth_xT(300 + c, dn * S_).reshape(S_, dn, 151) for c in range(b)])
    g = torch.Generator().manual_seed(77)
    t = torch.randint(0, 1000, (b,), generator=g)
    keep = torch.rand(b, generator=g) > 0.25
    seed = (2024, 1003)
    grads, sd = {}, None
    was = TE.TrainEngine.use_rows
    try:
        for rows in (False, True):
            TE.TrainEngine.use_rows = rows
            sd, diff = build("bf16", dn=dn, S_=S_, T_=1000)
            diff.train()
            diff.model.train_seed = seed
            total, _ = diff.p_losses(x_start.to(DEV), cond.to(DEV), t.to(DEV), noise=noise.to(DEV), keep_mask=keep.to(DEV))
            total.backward()
            assert sum(lk.use_rows for lk in diff.model.train_engine().lins.values()) == (56 if rows else 0)
            grads[rows] = {n: p.grad.cpu().numpy().copy() for n, p in diff.model.named_parameters() if p.grad is not None}
    finally:
        TE.TrainEngine.use_rows = was
    if ("c5", b) not in _ORACLE_CACHE:
        sd_now = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in sd.items() if p.is_floating_point()}
        o_total, o_losses = O.p_losses(sd_now, O.make_tables(1000), x_start, cond, t, noise, keep, drop=O.DropPlan(seed, 0.1))
        o_total.backward()
        _ORACLE_CACHE[("c5", b)] = (float(o_total), np.array([float(v) for v in o_losses]),
                                    {n: (None if v.grad is None else v.grad.numpy().copy()) for n, v in sd_now.items()})
    ograd = _ORACLE_CACHE[("c5", b)][2]
    tab = sorted(((rel(grads[True][n], ograd[n]), rel(grads[False][n], ograd[n]), rel(grads[True][n], grads[False][n]), n)
                  for n in grads[True]), reverse=True)
    med = [float(np.median([r[i] for r in tab])) for i in range(3)]
    print("[bf16] rows path vs oracle | tile path vs oracle | rows vs tile, rel-L2 per parameter: median "
          f"{med[0]:.2e} | {med[1]:.2e} | {med[2]:.2e}; worst by the rows path: " +
          ", ".join(f"{n} {a:.2e} | {c:.2e} | {d:.2e}" for a, c, d, n in tab[:5]))
    # The two paths differ from each other by about as much as either differs from the exact gradient: the bf16 step's error is
    # one draw of forward rounding noise (shared by all parameters through the perturbed output, so a draw moves every
    # parameter's error together), and a different summation order is another draw -- observed here 6.3e-2 (rows) / 1.6e-1
    # (tile) median, 1.3e-1 between them; at 32 clips the order is reversed (8.1e-2 / 5e-2).  Neither path is the accurate one.
    assert med[2] <= 1.5 * max(med[0], med[1])
    assert tab[0][0] < 4e-1 and max(r[1] for r in tab) < 4e-1


@pytest.mark.parametrize("b", [3])        # (round 6: 32 clips are held by the per-parameter pin below, against nine draws instead of one)
def test_bf16_step_against_the_oracle_that_rounds_where_the_kernels_round(b):
    """VERDICT r4 #3.  The bf16 step cannot be held to the EXACT fp32 gradient tightly: rounding the forward's GEMM / attention
    operands to bf16 alone moves the gradients by 5e-2 .. 1.4e-1 (median over the parameters) and up to 2.6e-1 (self- /
    cross-attention w_qs / w_ks of the last layers), whatever the backward does -- and two roundings of the same step (another
    summation order) land that far from each other too.  What it CAN be held to is the oracle with the kernels' rounding points
    (oracle.operand_rounding: operands in the forward, T-typed activation gradients, stored GEMM outputs): this test computes, on
    the same weights / draws / dropout masks (train mode, the full four-term loss, 3 dancers x 150 frames),
        e_k = kernels vs exact,   e_A, e_D = two emulated roundings vs exact   (per parameter, relative L2)
    and requires the kernels' error distribution to sit inside the emulations' own: median and worst <= 2.5 x the larger emulated
    value.  A defect in a backward kernel (a wrong dS, a dropped term, bf16 where fp32 is claimed) adds to e_k and not to e_A / e_D.
    Measured on the build host (CPU emulation): b = 3: e_A median 1.4e-1 / worst 2.6e-1, e_D 6.6e-2 / 1.4e-1 (without dropout 7.7e-2 /
    1.7e-1 and 6.3e-2 / 2.2e-1); b = 32: e_A 5.1e-2 / 1.8e-1, e_D 5.6e-2 / 2.2e-1 -- the kernels: 6.3e-2 / 2.1e-1 (b = 3), 8.1e-2 / 3.0e-1
    (b = 32), i.e. the fixed bounds of this file (3e-1 at the C1 shape, 4e-1 here) are ~1.4 x the emulation's own worst case."""
    dn, S_ = 3, 150
    sd, diff = build("bf16", dn=dn, S_=S_, T_=1000)
    model = diff.model
    diff.train()
    x_start = torch.stack([O.synth_motion(300 + c, dn * S_).reshape(S_, dn, 151).permute(1, 0, 2) for c in range(b)])
    cond = torch.stack([O.synth_cond(300 + c, S_) for c in range(b)])
    noise = torch.stack([O.synth_xT(300 + c, dn * S_).reshape(S_, dn, 151) for c in range(b)])
    g = torch.Generator().manual_seed(77)
    t = torch.randint(0, 1000, (b,), generator=g)
    keep = torch.rand(b, generator=g) > 0.25
    seed = (2024, 1003)
    model.train_seed = seed
    total, _ = diff.p_losses(x_start.to(DEV), cond.to(DEV), t.to(DEV), noise=noise.to(DEV), keep_mask=keep.to(DEV))
    for p in model.parameters():
        p.grad = None
    total.backward()
    gk = {n: p.grad.cpu().numpy().copy() for n, p in model.named_parameters() if p.grad is not None}
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))

    def oracle_grads(mode):
        sd_now = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in sd.items() if p.is_floating_point()}
        if mode is None:
            o_total, o_losses = O.p_losses(sd_now, O.make_tables(1000), x_start, cond, t, noise, keep, drop=O.DropPlan(seed, 0.1))
            o_total.backward()
        else:
            with O.operand_rounding(**mode):
                o_total, o_losses = O.p_losses(sd_now, O.make_tables(1000), x_start, cond, t, noise, keep, drop=O.DropPlan(seed, 0.1))
                o_total.backward()
        return (float(o_total), np.array([float(v) for v in o_losses]),
                {n: (None if v.grad is None else v.grad.numpy().copy()) for n, v in sd_now.items()})
    if ("c5", b) not in _ORACLE_CACHE:
        _ORACLE_CACHE[("c5", b)] = oracle_grads(None)
    g0 = _ORACLE_CACHE[("c5", b)][2]
    modes = {"A (forward operands)": dict(fwd=True, bwd=False), "D (operands, gradients, stored outputs)": dict(fwd=True, bwd=True, out=True)}
    if b == 32:
        modes.pop("A (forward operands)")       # one emulated evaluation at the large batch (a minute of CPU)
    stats = {}
    for name, mode in modes.items():
        ge = oracle_grads(mode)[2]
        e = np.array(sorted(rel(ge[n], g0[n]) for n in gk))
        ek_e = np.array(sorted(rel(gk[n], ge[n]) for n in gk))
        stats[name] = (float(np.median(e)), float(e[-1]))
        print(f"[bf16, {b} x 3 x 150] emulation {name} vs exact: median {np.median(e):.2e} worst {e[-1]:.2e}; kernels vs this emulation: "
              f"median {np.median(ek_e):.2e} worst {ek_e[-1]:.2e}")
    ek = sorted(((rel(gk[n], g0[n]), n) for n in gk), reverse=True)
    med_k, worst_k = float(np.median([v for v, _ in ek])), ek[0][0]
    med_e, worst_e = max(v[0] for v in stats.values()), max(v[1] for v in stats.values())
    print(f"[bf16, {b} x 3 x 150] kernels vs exact: median {med_k:.2e} worst {worst_k:.2e} ({ek[0][1]}); emulations: median <= {med_e:.2e}, "
          f"worst <= {worst_e:.2e}")
    # 2.5 x: two roundings of one step differ from each other by up to that much -- emulations A / D 1.4e-1 / 6.6e-2 (b = 3, build host),
    # the two HIP paths 6.3e-2 / 1.6e-1 (b = 3) -- and the emulated draw itself moves with the CPU's thread count; observed here on
    # MI355X boxes: kernels / emulation = 0.66 (b = 3), 1.76 (b = 32, median and worst alike)
    assert med_k <= 2.5 * med_e and worst_k <= 2.5 * worst_e, (med_k, med_e, worst_k, worst_e)


@pytest.mark.parametrize("b", [3, 32])
def test_bf16_step_error_profile_is_a_draw_of_the_emulated_rounding_noise(golden_dir, b):
    """the PIN.  tests/golden/c5_bf16_draws.npz (make_golden_bf16_draws.py, the CPU oracle alone) holds nine draws of the
    oracle evaluated with the kernels' rounding points -- each perturbed immaterially (x_start x (1 + k 2^-18)) -- as per-parameter
    relative L2 distances to the exact fp32 gradient.  What they show: (i) the SCALE of the error is a property of the draw (median over
    the parameters 5.4e-2 .. 1.4e-1 at 3 clips, 4.3e-2 .. 8.3e-2 at 32 clips; worst parameter 1.5e-1 .. 2.7e-1 and 1.6e-1 .. 3.1e-1): round
    5's "kernels / emulation = 1.76 at 32 clips" compared the kernels with ONE draw (4.6e-2) of a quantity whose own draws span 1.9 x;
    (ii) the SHAPE of the error -- a parameter's error divided by the draw's median -- is the same in every draw: leave-one-out, no
    parameter of any draw exceeds the envelope of the other eight by more than 1.15 x (which parameters suffer, and by how much relative
    to the rest, is decided by the loss, not by the draw).
    So the HIP step is held to both, per parameter: every one of the ~308 live parameters' (error / median error) within SHAPE x the
    draws' envelope -- a defect in one backward kernel (attention dS, row_bwd's d_z, an input-gradient GEMM) raises ITS parameters
    against the rest and fails here, where the old distribution envelope (median / worst <= 2.5 x one emulation) let 40 % through --
    and the median / worst within the draws' range."""
    z = np.load(os.path.join(golden_dir, "c5_bf16_draws.npz"))
    names, E = [str(n) for n in z[f"b{b}_names"]], z[f"b{b}_err"].astype(np.float64)
    gk, g0 = _c5_bf16_step(b)
    idx = [i for i, n in enumerate(names) if n in gk and E[:, i].max() > 0]
    assert len(idx) >= 300, len(idx)
    names, E = [names[i] for i in idx], E[:, idx]
    ek = np.array([rel(gk[n], g0[n]) for n in names])
    med_d, worst_d = np.median(E, axis=1), E.max(axis=1)
    med_k, worst_k = float(np.median(ek)), float(ek.max())
    shape_hi = (E / med_d[:, None]).max(axis=0)
    ratio = (ek / med_k) / shape_hi
    order = np.argsort(-ratio)
    print(f"[bf16, {b} x 3 x 150] kernels vs exact: median {med_k:.2e} worst {worst_k:.2e} ({names[int(ek.argmax())]}); the nine emulated "
          f"draws: median {med_d.min():.2e} .. {med_d.max():.2e}, worst {worst_d.min():.2e} .. {worst_d.max():.2e}; error SHAPE against the draws' "
          f"envelope: max {ratio[order[0]]:.2f} ({names[order[0]]}), then " + ", ".join(f"{ratio[i]:.2f} {names[i]}" for i in order[1:4]) +
          f"; 95th percentile {np.percentile(ratio, 95):.2f}")
    # scale: inside the draws' own range (a quarter above its top: nine draws do not exhaust a distribution)
    assert 0.5 * med_d.min() <= med_k <= 1.25 * med_d.max(), (med_k, med_d)
    assert worst_k <= 1.25 * worst_d.max(), (worst_k, worst_d)
    # shape: leave-one-out the draws need 1.15; the kernels add their own summation orders (row-block GEMM streams, fp32 atomics)
    assert ratio.max() <= SHAPE_TOL, (names[order[0]], float(ratio.max()))



def _c5_bf16_step(b):
    """One bf16 train-mode step of the HIP kernels at b x 3 x 150 and the oracle's exact fp32 gradients on the same inputs."""
    if ("k", b) in _ORACLE_CACHE:
        return _ORACLE_CACHE[("k", b)]
    dn, S_ = 3, 150
    sd, diff = build("bf16", dn=dn, S_=S_, T_=1000)
    model = diff.model
    diff.train()
    x_start = torch.stack([O.synth_motion(300 + c, dn * S_).reshape(S_, dn, 151).permute(1, 0, 2) for c in range(b)])
    cond = torch.stack([O.synth_cond(300 + c, S_) for c in range(b)])
    noise = torch.stack([O.synth_xT(300 + c, dn * S_).reshape(S_, dn, 151) for c in range(b)])
    g = torch.Generator().manual_seed(77)
    t = torch.randint(0, 1000, (b,), generator=g)
    keep = torch.rand(b, generator=g) > 0.25
    seed = (2024, 1003)
    model.train_seed = seed
    total, _ = diff.p_losses(x_start.to(DEV), cond.to(DEV), t.to(DEV), noise=noise.to(DEV), keep_mask=keep.to(DEV))
    for p in model.parameters():
        p.grad = None
    total.backward()
    gk = {n: p.grad.cpu().numpy().copy() for n, p in model.named_parameters() if p.grad is not None}
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    if ("c5", b) not in _ORACLE_CACHE:
        sd_now = {n: p.detach().cpu().clone().requires_grad_(True) for n, p in sd.items() if p.is_floating_point()}
        o_total, o_losses = O.p_losses(sd_now, O.make_tables(1000), x_start, cond, t, noise, keep, drop=O.DropPlan(seed, 0.1))
        o_total.backward()
        _ORACLE_CACHE[("c5", b)] = (float(o_total), np.array([float(v) for v in o_losses]),
                                   {n: (None if v.grad is None else v.grad.numpy().copy()) for n, v in sd_now.items()})
    _ORACLE_CACHE[("k", b)] = (gk, _ORACLE_CACHE[("c5", b)][2])
    return _ORACLE_CACHE[("k", b)]


_ORACLE_CACHE = {}


def K_tn_taken(eng, M):
    """the shape really takes tcdiff_gemm_tn (token rows a multiple of the k-tile, 128-wide operands)"""
    from tcdiff_amd import kernels as K
    return K.gemm_tn_ok(eng.dt, 512, 512, M) and K.gemm_tn_ok(eng.dt, 1024, 512, M)


def test_reference_training_loop_runs_unchanged():
    """TCDiff.train_loop's body (TCDiff.py:222-245) verbatim on the drop-in classes: .train(), diffusion(x, cond),
    zero_grad, backward, Adan.step, EMA -- random t / noise / keep mask / dropout seed drawn as the reference draws them."""
    torch.manual_seed(0)
    sd, diff = build("bf16")
    diff.train()
    optim = Adan(diff.model.parameters(), lr=5e-5, weight_decay=0.02)
    x_start, cond, _ = step_inputs(0, 10)
    before = {n: p.detach().clone() for n, p in diff.model.named_parameters()}
    losses = []
    for step in range(3):
        total_loss, (loss, v_loss, fk_loss, foot_loss) = diff(x_start.to(DEV), cond.to(DEV), t_override=None)
        optim.zero_grad()
        total_loss.backward()
        optim.step()
        diff.ema.update_model_average(diff.master_model, diff.model)
        losses.append(float(total_loss))
        assert all(bool(torch.isfinite(v)) for v in (total_loss, loss, v_loss, fk_loss, foot_loss))
    moved = sum(int(not torch.equal(p.detach(), before[n])) for n, p in diff.model.named_parameters())
    assert moved == 435 - 125, moved                # every live parameter was updated; the 125 unused ones have no gradient
    # gradient accumulation without zero_grad: the second backward must ADD to .grad (torch semantics), not alias it
    diff.model.train_seed = (7, 7)
    t = torch.tensor([3, 30, 60], device=DEV)
    optim.zero_grad()
    tot, _ = diff.p_losses(x_start.to(DEV), cond.to(DEV), t, noise=torch.zeros(B, S, DN, 151, device=DEV),
                           keep_mask=torch.tensor([True, True, False], device=DEV))
    tot.backward()
    p = diff.model.final_layer.weight
    g1 = p.grad.detach().clone()
    diff.model.train_seed = (7, 7)
    tot, _ = diff.p_losses(x_start.to(DEV), cond.to(DEV), t, noise=torch.zeros(B, S, DN, 151, device=DEV),
                           keep_mask=torch.tensor([True, True, False], device=DEV))
    tot.backward()
    assert float((p.grad - 2 * g1).abs().max()) <= 1e-3 * float(g1.abs().max())


@pytest.mark.parametrize("mode", [2, 1])
def test_captured_step_equals_the_eager_schedule(mode):
    """From the third step with the same shapes on, forward and backward are replayed (train_engine.py: 2 = the captured
    hipGraphs, 1 = the recorded launcher calls).  Same launches, so the same numbers: the forward bit for bit, the gradients up
    to the summation order of the fp32 atomics -- checked step by step against the eager schedule on the same parameters."""
    x_start, cond, _ = step_inputs(0, 10)
    torch.manual_seed(5)
    sd, diff = build("f32")
    diff.train()
    eng = diff.model.train_engine()
    optim = Adan(diff.model.parameters(), lr=1e-4, weight_decay=0.0)
    named = dict(diff.model.named_parameters())

    def grads(graphs, step):
        eng.use_graphs = mode if graphs else 0
        diff.model.train_seed = (11, step)
        t = torch.tensor([3 + step, 30, 60 + step], device=DEV)
        tot, _ = diff.p_losses(x_start.to(DEV), cond.to(DEV), t, noise=torch.zeros(B, S, DN, 151, device=DEV),
                               keep_mask=torch.tensor([True, step % 2 == 0, False], device=DEV))
        optim.zero_grad()
        tot.backward()
        return float(tot), {n: p.grad.detach().clone() for n, p in named.items() if p.grad is not None}

    for step in range(6):
        le, ge = grads(False, step)
        lg, gg = grads(True, step)
        assert eng._graph_broken is None, eng._graph_broken
        assert le == lg, (step, le, lg)                                   # the forward has no atomics
        worst = max(rel(gg[n].cpu().numpy(), ge[n].cpu().numpy()) for n in ge)
        assert set(gg) == set(ge) and worst < 2e-5, (step, worst)
        optim.step()                                                      # on the captured path's gradients
    st = [v for v in eng._graphs.values() if v["fwd"] is not None and v["bwd"] is not None]
    assert len(st) == 1 and st[0]["n"] == 6


@pytest.mark.parametrize("compute", ["bf16", "f32"])
def test_training_on_a_fixed_batch_reduces_the_loss(compute):
    """end-to-end sanity beyond per-step gradient parity: 300 steps of the reference's loop (dropout live, random t / noise /
    conditioning drop, replayed graphs from step 3 on) on one fixed batch of smooth motion drive its loss down"""
    torch.manual_seed(1)
    sd, diff = build(compute)
    diff.train()
    optim = Adan(diff.model.parameters(), lr=1e-3, weight_decay=0.02)
    _, cond, _ = step_inputs(0, 10)
    g = torch.Generator().manual_seed(3)
    # a learnable target: smooth motion (1-3 cycles of a sinusoid per channel), not the white noise of the parity inputs whose
    # velocity term no model can fit
    frames = torch.arange(S, dtype=torch.float32).view(1, 1, S, 1) / S
    freq = torch.randint(1, 4, (B, DN, 1, 151), generator=g).float()
    x_start = 0.6 * torch.sin(2 * math.pi * (freq * frames + torch.rand(B, DN, 1, 151, generator=g)))
    x_start, cond = x_start.to(DEV), cond.to(DEV)
    hist = []
    for step in range(300):
        total, _ = diff(x_start, cond)
        optim.zero_grad()
        total.backward()
        optim.step()
        hist.append(total.detach())
    hist = torch.stack(hist).cpu()
    first, last = float(hist[:20].mean()), float(hist[-20:].mean())
    print(f"[{compute}] loss, mean of steps 0-19: {first:.4f}; of steps 280-299: {last:.4f}")
    assert bool(torch.isfinite(hist).all()) and last < 0.85 * first, (first, last)          # observed 0.68 (bf16), random t per step
    assert diff.model.train_engine()._graph_broken is None


def test_changing_batch_sizes_keep_at_most_three_captured_shapes():
    """every new batch shape runs eagerly twice, is captured at its third step, and the oldest captured shape is dropped when a
    fourth arrives (a captured shape pins its activations); going back to an evicted shape simply starts over"""
    torch.manual_seed(2)
    sd, diff = build("bf16")
    diff.train()
    eng = diff.model.train_engine()
    optim = Adan(diff.model.parameters(), lr=1e-4, weight_decay=0.0)
    for b in (1, 2, 3, 4, 1):
        x = torch.stack([O.synth_motion(c, DN * S).reshape(S, DN, 151).permute(1, 0, 2) for c in range(b)]).to(DEV)
        cond = torch.stack([O.synth_cond(c, S) for c in range(b)]).to(DEV)
        for _ in range(4):
            total, _ = diff(x, cond)
            optim.zero_grad()
            total.backward()
            optim.step()
            assert bool(torch.isfinite(total))
        captured = [k for k, v in eng._graphs.items() if v.get("fwd") is not None]
        assert eng._graph_broken is None and 1 <= len(captured) <= 3 and captured[-1][0] == b, (b, captured)


def test_inference_after_training_sees_the_updated_weights():
    sd, diff = build("f32")
    diff.train()
    optim = Adan(diff.model.parameters(), lr=1e-3, weight_decay=0.0)
    x_start, cond, noise = step_inputs(0, 10)
    for _ in range(2):
        tot, _ = diff(x_start.to(DEV), cond.to(DEV))
        optim.zero_grad()
        tot.backward()
        optim.step()
    diff.eval()
    x = torch.stack([O.synth_xT(c, DN * S) for c in range(B)])
    t = torch.tensor([10, 10, 10])
    with torch.no_grad():
        got = diff.model(x.to(DEV), cond.to(DEV), t.to(DEV))
        sd_now = {n: p.detach().cpu() for n, p in diff.model.state_dict().items()}
        want = O.decoder_forward(sd_now, x, cond, t)
    assert float((got.cpu() - want).abs().max()) < 5e-4


def _fixed_step_inputs():
    x_start, cond, noise = step_inputs(0, 10)
    t = torch.tensor([73, 5, 40])
    keep = torch.tensor([True, False, True])
    return [v.to(DEV) for v in (x_start, cond, t, noise, keep)]


def test_backward_of_a_stale_forward_raises_instead_of_using_the_newer_activations():
    """the engine keeps the activations (and dropout seed) of ONE forward.  fwd(A), fwd(B), loss_A.backward() used
    to differentiate A through B's activations silently; the autograd node now carries the generation of its forward."""
    from tcdiff_amd._lib import TcdiffError
    _, diff = build("f32")
    diff.eval()
    x, cond, t, noise, keep = _fixed_step_inputs()
    la, _ = diff.p_losses(x, cond, t, noise=noise, keep_mask=keep)
    lb, _ = diff.p_losses(x * 0.5, cond, t, noise=noise, keep_mask=keep)
    with pytest.raises(TcdiffError, match="no longer the most recent"):
        la.backward()
    lb.backward()                                   # the most recent forward is still differentiable
    assert diff.model.final_layer.weight.grad is not None


def test_reallocated_parameters_drop_the_captured_graphs_and_frozen_ones_get_no_gradient():
    """captured forward / backward graphs bake in raw parameter addresses (LayerNorm weights, biases, the weight-pack
    table).  After a parameter's storage is replaced (`p.data = ...`) the next steps must recapture and still equal the eager
    schedule; a parameter with requires_grad=False receives no .grad; `grads_through_autograd` hands gradients to autograd (a
    tensor hook on a parameter fires)."""
    _, diff = build("f32")
    _, ref = build("f32")
    for d in (diff, ref):
        d.eval()
    ref.model.train_engine().use_graphs = 0
    x, cond, t, noise, keep = _fixed_step_inputs()

    def grads(d, steps):
        out = None
        for _ in range(steps):
            for p in d.model.parameters():
                p.grad = None
            total, _ = d.p_losses(x, cond, t, noise=noise, keep_mask=keep)
            total.backward()
            out = {n: p.grad.detach().clone() for n, p in d.model.named_parameters() if p.grad is not None}
        return out
    g_ref = grads(ref, 1)
    g0 = grads(diff, 4)                             # eager, eager, captured, replayed
    eng = diff.model.train_engine()
    assert any(st["fwd"] is not None for st in eng._graphs.values())
    assert max(rel(g0[n].cpu().numpy(), g_ref[n].cpu().numpy()) for n in g_ref) < 1e-5
    # move two parameters to new storage (what load_state_dict(assign=True) / model.to() on the same device do) and change them
    with torch.no_grad():
        for name in ("seqTransDecoder.stack.3.norm2.weight", "final_layer.bias"):
            p = dict(diff.model.named_parameters())[name]
            p.data = (p.data * 1.5).clone()
            q = dict(ref.model.named_parameters())[name]
            q.mul_(1.5)
    g1, g1_ref = grads(diff, 4), grads(ref, 1)
    assert eng._graph_broken is None
    assert max(rel(g1[n].cpu().numpy(), g1_ref[n].cpu().numpy()) for n in g1_ref) < 1e-5
    assert rel(g1["final_layer.weight"].cpu().numpy(), g0["final_layer.weight"].cpu().numpy()) > 1e-6    # it did change
    # frozen parameter: no gradient; hook on a live one with grads_through_autograd
    frozen = diff.model.seqTransDecoder.stack[0].linear1.weight
    frozen.requires_grad_(False)
    seen = []
    live = diff.model.final_layer.weight
    h = live.register_hook(lambda g: seen.append(float(g.abs().sum())))
    g2 = grads(diff, 1)
    h.remove()
    assert "seqTransDecoder.stack.0.linear1.weight" not in g2 and len(seen) == 1 and seen[0] > 0
    eng.grads_through_autograd = True
    g3 = grads(diff, 1)
    assert "seqTransDecoder.stack.0.linear1.weight" not in g3
    assert rel(g3["final_layer.weight"].cpu().numpy(), g2["final_layer.weight"].cpu().numpy()) < 1e-6
