"""End-to-end parity on MI355X: tcdiff_amd (HIP kernels through the C ABI) against
  (a) the golden vectors produced by the real reference (tests/golden, committed), and
  (b) the CPU oracle (oracle/tcdiff_oracle.py) run here on the same seeded inputs.

Tolerances: BASELINE.json north_star asks max-abs <= 1e-3 in fp32 for the same seed / noise schedule.  The f32
mode (v_mfma_f32_32x32x2_f32, exact fp32 fma chains) is held to 1e-3 end-to-end and to 2e-4 per evaluation; the
bf16 mode (throughput) is reported against the oracle with its own, looser, stated bound."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import tcdiff_oracle as O  # noqa: E402  (checker only)
from tcdiff_amd.diffusion import GaussianDiffusion  # noqa: E402
from tcdiff_amd.model import DanceDecoder  # noqa: E402

DEV = "cuda"
torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))


def build(dn, S, T, compute="f32"):
    sd = O.synth_state_dict(dn=dn, seq_len=S)
    model = DanceDecoder(nfeats=151, seq_len=S, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                         cond_feature_dim=438, activation=F.gelu, required_dancer_num=dn, compute_dtype=compute)
    model.load_state_dict(sd, strict=True)
    model.eval()
    diff = GaussianDiffusion(model, S, 151, None, schedule="cosine", n_timestep=T, predict_epsilon=False,
                             loss_type="l2", use_p2=False, cond_drop_prob=0.25, guidance_weight=2, seq_len=S)
    diff.to(DEV).eval()
    return sd, model, diff


def gold(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def maxabs(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, dtype=np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)))


def dev_noise(clip_ids, Lq):
    fn = O.batch_step_noise(clip_ids, Lq)
    return lambda t, shape: fn(t, shape).to(DEV)


@pytest.fixture(scope="module")
def c1():
    sd, model, diff = build(2, 60, 100)
    cond = torch.stack([O.synth_cond(0, 60)])
    xT = torch.stack([O.synth_xT(0, 120)])
    return sd, model, diff, cond, xT


def test_tables_match_reference(golden_dir, c1):
    _, _, diff, _, _ = c1
    # The tables are evaluated by the HOST's libm/SIMD paths: bit-exact against the reference in the build
    # container (tests/test_host_cpu.py, tests/test_oracle_golden.py); on another CPU model the last bit of a few
    # entries can differ, so here the bound is 2 ulp of fp32.
    ref = gold(golden_dir, "tables_T100")
    for k in ref.files:
        got = getattr(diff, k).cpu().numpy()
        err = np.max(np.abs(got.astype(np.float64) - ref[k]) / (np.abs(ref[k]) + 1e-30))
        assert err < 2.5e-7, (k, err)


def test_c1_forward_vs_reference_golden(golden_dir, c1):
    _, model, _, cond, xT = c1
    ref = gold(golden_dir, "c1_forward")
    for t in (99, 3):
        tt = torch.full((1,), t, dtype=torch.long, device=DEV)
        e_c = maxabs(model(xT.to(DEV), cond.to(DEV), tt, cond_drop_prob=0.0), ref[f"fwd_cond_t{t}"])
        e_u = maxabs(model(xT.to(DEV), cond.to(DEV), tt, cond_drop_prob=1.0), ref[f"fwd_unc_t{t}"])
        e_g = maxabs(model.guided_forward(xT.to(DEV), cond.to(DEV), tt, 2), ref[f"guided_w2_t{t}"])
        print(f"C1 forward t={t}: cond {e_c:.2e} unc {e_u:.2e} guided {e_g:.2e}")
        assert e_c < 2e-4 and e_u < 2e-4 and e_g < 4e-4


def test_c1_p_sample_loop_vs_reference_golden(golden_dir, c1):
    """BASELINE config 1 end to end: 1 clip, 2 dancers x 60 frames, 100 DDPM steps, same injected noise."""
    _, _, diff, cond, xT = c1
    ref = gold(golden_dir, "c1_p_sample_loop")
    x, chain = diff.p_sample_loop((1, 120, 151), cond, noise=xT, step_noise=dev_noise([0], 120), return_diffusion=True)
    errs = {k: maxabs(chain[i], ref[k]) for k, i in (("after_step_99", 1), ("after_step_50", 50), ("after_step_10", 90),
                                                    ("after_step_1", 99))}
    errs["final"] = maxabs(x, ref["final"])
    print("C1 p_sample_loop max-abs vs reference:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) < 1e-3


@pytest.mark.parametrize("name,eps,clip,start", [("loop_eps_clip", True, True, 30), ("loop_eps_noclip", True, False, 12),
                                                 ("loop_x0_noclip", False, False, 30)])
def test_constructor_options_outside_the_production_config_vs_reference_golden(golden_dir, name, eps, clip, start):
    """predict_epsilon=True (the reference constructor's DEFAULT) and clip_denoised=False (model/diffusion.py:80-95,176-187,
    230-233), which rounds 1-3 refused: the last `start` DDPM steps of config 1 against the REAL reference with the same
    injected noise (tests/golden/make_golden_eps.py), f32 mode, 1e-3 of the largest sample value."""
    ref = gold(golden_dir, "c1_eps")[name]
    sd = O.synth_state_dict(dn=2, seq_len=60)
    model = DanceDecoder(nfeats=151, seq_len=60, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                         cond_feature_dim=438, activation=F.gelu, required_dancer_num=2, compute_dtype="f32")
    model.load_state_dict(sd, strict=True)
    diff = GaussianDiffusion(model.eval(), 60, 151, None, schedule="cosine", n_timestep=100, predict_epsilon=eps,
                             clip_denoised=clip, loss_type="l2", use_p2=False, cond_drop_prob=0.25, guidance_weight=2,
                             seq_len=60).to(DEV).eval()
    cond = torch.stack([O.synth_cond(0, 60)])
    xT = torch.stack([O.synth_xT(0, 120)])
    x = diff.p_sample_loop((1, 120, 151), cond, noise=xT, step_noise=dev_noise([0], 120), start_point=start)
    e, scale = maxabs(x, ref), float(np.abs(ref).max())
    print(f"{name}: max-abs vs reference {e:.2e} (|x| max {scale:.2f})")
    assert e < 1e-3 * max(1.0, scale)
    # p_sample's second return value is x_start of that step (:241-252)
    tt = torch.full((1,), 5, dtype=torch.long, device=DEV)
    xn, x0 = diff.p_sample(xT.to(DEV), cond.to(DEV), tt, noise=torch.zeros_like(xT).to(DEV))
    assert bool(torch.isfinite(x0).all()) and (not clip or float(x0.abs().max()) <= 1.0)


@pytest.mark.parametrize("name,clip", [("noclip", False), ("clip", True)])
def test_ddim_sample_honours_clip_denoised_vs_reference_golden(golden_dir, name, clip):
    """The DDIM samplers pass clip_x_start=self.clip_denoised (model/diffusion.py:316,409,476): 50 DDIM steps of config 1
    against the REAL reference with and without the clamp, same injected draws (tests/golden/make_golden_ddim_noclip.py; the
    two goldens differ by 0.65, so a sampler that always clamps fails the first), f32 mode."""
    ref = gold(golden_dir, "c1_ddim_noclip")[name]
    sd = O.synth_state_dict(dn=2, seq_len=60)
    model = DanceDecoder(nfeats=151, seq_len=60, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                         cond_feature_dim=438, activation=F.gelu, required_dancer_num=2, compute_dtype="f32")
    model.load_state_dict(sd, strict=True)
    diff = GaussianDiffusion(model.eval(), 60, 151, None, schedule="cosine", n_timestep=100, predict_epsilon=False,
                             clip_denoised=clip, loss_type="l2", use_p2=False, cond_drop_prob=0.25, guidance_weight=2,
                             seq_len=60).to(DEV).eval()
    cond = torch.stack([O.synth_cond(0, 60)])
    xT = torch.stack([O.synth_xT(0, 120)])
    x = diff.ddim_sample((1, 120, 151), cond, init_noise=xT, step_noise=dev_noise([0], 120))
    e, scale = maxabs(x, ref), float(np.abs(ref).max())
    print(f"ddim_sample clip_denoised={clip}: max-abs vs reference {e:.2e} (|x| max {scale:.2f})")
    assert e < 1e-3 * max(1.0, scale)
    # p_mean_variance is the same arithmetic as p_sample's mean (model/diffusion.py:215-239)
    tt = torch.full((1,), 5, dtype=torch.long, device=DEV)
    mean, _, _, x0 = diff.p_mean_variance(xT.to(DEV), cond.to(DEV), tt)
    xn, x0b = diff.p_sample(xT.to(DEV), cond.to(DEV), tt, noise=torch.zeros_like(xT).to(DEV))
    assert maxabs(mean, xn.cpu().numpy()) < 1e-5 and maxabs(x0, x0b.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("name", ["relu", "silu", "mish"])
def test_feed_forward_activation_option_vs_reference_golden(golden_dir, name):
    """DanceDecoder(activation=F.relu | F.silu | F.mish) (model/model.py:244,400; rounds 1-4 refused everything but F.gelu): a guided
    evaluation and a conditional forward of config 1 against the REAL reference built with that activation
    (tests/golden/make_golden_activation.py).  f32 mode to 1e-3; the bf16 mode runs these on the op-by-op kernels (the fused chain
    kernels are GELU only) within its stated bound."""
    ref = gold(golden_dir, "c1_activation")
    fn = {"relu": F.relu, "silu": F.silu, "mish": F.mish}[name]
    sd = O.synth_state_dict(dn=2, seq_len=60)
    cond = torch.stack([O.synth_cond(0, 60)]).to(DEV)
    xT = torch.stack([O.synth_xT(0, 120)]).to(DEV)
    for compute, bound in (("f32", 1e-3), ("bf16", BF16_EVAL_BOUND)):
        model = DanceDecoder(nfeats=151, seq_len=60, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                             cond_feature_dim=438, activation=fn, required_dancer_num=2, compute_dtype=compute).to(DEV).eval()
        model.load_state_dict(sd, strict=True)
        g = model.guided_forward(xT, cond, torch.full((1,), 50, dtype=torch.long, device=DEV), 2)
        c = model(xT, cond, torch.full((1,), 3, dtype=torch.long, device=DEV), cond_drop_prob=0.0)
        e1, e2 = maxabs(g, ref[f"{name}_guided_w2_t50"]), maxabs(c, ref[f"{name}_fwd_cond_t3"])
        print(f"activation={name} [{compute}]: guided t=50 {e1:.2e}, conditional t=3 {e2:.2e} (bound {bound})")
        assert e1 < bound and e2 < bound
        assert not model.engine(1).use_chain


def test_use_rotary_false_option_vs_reference_golden(golden_dir):
    """DanceDecoder(use_rotary=False) (model/model.py:441-448: no rotary embedding, PositionalEncoding added to the motion tokens :564
    and the music tokens :580; rounds 1-4 refused it): a guided evaluation, a conditional forward and the full 100-step p_sample_loop
    of config 1 against the REAL reference built with that option (tests/golden/make_golden_abs_pos.py), on the
    op-by-op kernels (identity rotary table, positional rows added by tcdiff_add_rows); f32 mode to 1e-3, bf16 within its bound."""
    from tcdiff_amd import _lib as L
    ref = gold(golden_dir, "c1_abs_pos")
    sd = O.synth_state_dict(dn=2, seq_len=60, use_rotary=False)
    cond = torch.stack([O.synth_cond(0, 60)]).to(DEV)
    xT = torch.stack([O.synth_xT(0, 120)]).to(DEV)
    for compute, bound, lbound in (("f32", 1e-3, 1e-3), ("bf16x3", 1e-3, 1e-3), ("bf16", BF16_EVAL_BOUND, BF16_STEPS_BOUND)):
        model = DanceDecoder(nfeats=151, seq_len=60, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                             cond_feature_dim=438, activation=F.gelu, required_dancer_num=2, use_rotary=False,
                             compute_dtype=compute).to(DEV).eval()
        model.load_state_dict(sd, strict=True)
        diff = GaussianDiffusion(model, 60, 151, None, schedule="cosine", n_timestep=100, predict_epsilon=False, loss_type="l2",
                                 use_p2=False, cond_drop_prob=0.25, guidance_weight=2, seq_len=60)
        diff.to(DEV).eval()
        with torch.no_grad():
            g = model.guided_forward(xT, cond, torch.full((1,), 50, dtype=torch.long, device=DEV), 2)
            c = model(xT, cond, torch.full((1,), 3, dtype=torch.long, device=DEV), cond_drop_prob=0.0)
        e1, e2 = maxabs(g, ref["guided_w2_t50"]), maxabs(c, ref["fwd_cond_t3"])
        x, chain = diff.p_sample_loop((1, 120, 151), cond, noise=xT, step_noise=dev_noise([0], 120), return_diffusion=True)
        errs = {k: maxabs(chain[i], ref[k]) for k, i in (("after_step_99", 1), ("after_step_50", 50), ("after_step_10", 90),
                                                        ("after_step_1", 99))}
        errs["final"] = maxabs(x, ref["final"])
        print(f"use_rotary=False [{compute}]: guided t=50 {e1:.2e}, conditional t=3 {e2:.2e} (bound {bound}); loop",
              {k: f"{v:.2e}" for k, v in errs.items()}, f"(bound {lbound})")
        assert e1 < bound and e2 < bound and max(errs.values()) < lbound
        assert not model.engine(1).use_chain and model.engine(1).abs_pos
        # with gradients enabled the same call is the training path (eval mode: dropout off): the same values, now with a graph
        # (its gradients: tests/test_train_step_gpu.py::test_training_step_without_rotary_embedding_vs_oracle_autograd)
        model.eval()
        y = model(xT, cond, torch.full((1,), 3, dtype=torch.long, device=DEV), cond_drop_prob=0.0)
        assert y.requires_grad and maxabs(y.detach(), ref["fwd_cond_t3"]) < bound


def test_c1_graph_and_eager_agree(c1):
    _, _, diff, cond, xT = c1
    a = diff.p_sample_loop((1, 120, 151), cond, noise=xT, step_noise=dev_noise([0], 120), start_point=12, use_graph=True)
    b = diff.p_sample_loop((1, 120, 151), cond, noise=xT, step_noise=dev_noise([0], 120), start_point=12, use_graph=False)
    assert torch.equal(a, b)


@pytest.fixture(scope="module")
def c2():
    sd, model, diff = build(3, 150, 1000)
    cond = torch.stack([O.synth_cond(c, 150) for c in (0, 1)])
    xT = torch.stack([O.synth_xT(c, 450) for c in (0, 1)])
    return sd, model, diff, cond, xT


def test_c2_forward_vs_reference_golden(golden_dir, c2):
    _, model, _, cond, xT = c2
    ref = gold(golden_dir, "c2_forward")
    for t in (999, 37):
        tt = torch.full((1,), t, dtype=torch.long, device=DEV)
        e = maxabs(model.guided_forward(xT[:1].to(DEV), cond[:1].to(DEV), tt, 2), ref[f"guided_w2_t{t}"])
        print(f"C2 guided t={t}: {e:.2e}")
        assert e < 4e-4
    out = model(xT.to(DEV), cond.to(DEV), torch.tensor([500, 20], device=DEV), cond_drop_prob=0.0)
    e = maxabs(out, ref["fwd_cond_b2_t500_20"])
    print(f"C2 forward per-clip timesteps: {e:.2e}")
    assert e < 2e-4


def test_c2_ddpm_steps_vs_reference_golden(golden_dir, c2):
    _, _, diff, cond, xT = c2
    ref = gold(golden_dir, "c2_ddpm_steps")
    from tcdiff_amd import _lib as L
    tseq = [999, 998, 997]
    chain = []
    diff._run(L.SAMPLER_DDPM, (1, 450, 151), cond[:1], xT[:1].to(DEV), tseq, diff._ddpm_params(tseq),
              step_noise=dev_noise([0], 450), collect=chain)
    for j, i in enumerate(tseq):
        e = maxabs(chain[j], ref[f"after_step_{i}"])
        print(f"C2 DDPM step {i}: {e:.2e}")
        assert e < 5e-4


def test_c2_ddpm_low_t_steps(golden_dir, c2):
    _, _, diff, cond, xT = c2
    ref = gold(golden_dir, "c2_ddpm_steps")
    x, chain = diff.p_sample_loop((1, 450, 151), cond[:1], noise=xT[:1], start_point=3, step_noise=dev_noise([0], 450),
                                  return_diffusion=True)
    for j, i in enumerate((2, 1, 0)):
        e = maxabs(chain[j + 1], ref[f"after_step_{i}"])
        print(f"C2 DDPM step {i} (w clipped to 1): {e:.2e}")
        assert e < 5e-4


def test_c2_ddim_with_trajectory_vs_reference_golden(golden_dir, c2):
    _, _, diff, cond, xT = c2
    ref = gold(golden_dir, "c2_ddim")
    x0 = torch.stack([O.synth_traj(0, 450)])
    x = diff.ddim_sample((1, 450, 151), cond[:1], x_0=x0, init_noise=xT[:1], step_noise=dev_noise([0], 450))
    e = maxabs(x, ref["final"])
    print(f"C2 ddim_sample (50 steps, trajectory in-painting): {e:.2e}")
    assert e < 1e-3


def test_c2_long_ddim_vs_reference_golden(golden_dir, c2):
    _, _, diff, cond, xT = c2
    ref = gold(golden_dir, "c2_long_ddim")
    x0 = torch.stack([O.synth_traj(c, 450) for c in (0, 1)]).reshape(2, 150, 3, 3)
    x = diff.long_ddim_sample((2, 450, 151), cond, x0, init_noise=xT, step_noise=dev_noise([0, 1], 450))
    e = maxabs(x, ref["final"])
    print(f"C2 long_ddim_sample (window coupling + weight ramp): {e:.2e}")
    assert e < 1e-3


def test_c2_footwork_ddim_vs_reference_golden(golden_dir, c2):
    """ddim_sample_Footwork (model/diffusion.py:289-383): trajectory + lower-body rotations of frames 75:120 re-imposed
    every step, 10-frame blend at the end."""
    _, _, diff, cond, xT = c2
    ref = gold(golden_dir, "c2_footwork")
    x0 = torch.stack([O.synth_motion(0, 450)])
    x = diff.ddim_sample_Footwork((1, 450, 151), cond[:1], x_0=x0, init_noise=xT[:1], step_noise=dev_noise([0], 450))
    e = maxabs(x, ref["final"])
    print(f"C2 ddim_sample_Footwork (50 steps): {e:.2e}")
    assert e < 1e-3


def test_c1_inpaint_loop_vs_reference_golden(golden_dir, c1):
    """inpaint_loop (model/diffusion.py:519-557), 100 DDPM steps with the constraint re-imposed through q_sample."""
    _, _, diff, cond, xT = c1
    ref = gold(golden_dir, "c1_inpaint")
    value = torch.stack([O.synth_motion(0, 120)]).to(DEV)
    mask = torch.stack([O.synth_inpaint_mask(120)]).to(DEV)
    x = diff.inpaint_loop((1, 120, 151), cond, noise=xT, constraint={"mask": mask, "value": value},
                          step_noise=dev_noise([0], 120),
                          q_noise=lambda t, shape: torch.stack([O.synth_q_eps(0, t, 120)]))
    e = maxabs(x, ref["final"])
    print(f"C1 inpaint_loop (100 steps): {e:.2e}")
    assert e < 1e-3


def test_c1_inpaint_loop_with_predict_epsilon_vs_reference_golden(golden_dir):
    """inpaint_loop with predict_epsilon=True (the reference constructor's default; rounds 1-4 refused the combination: the
    constraint kernel now has its own step table): the last 30 DDPM steps against the REAL reference
    (tests/golden/make_golden_inpaint_eps.py), every draw injected."""
    ref = gold(golden_dir, "c1_inpaint_eps")["final"]
    sd = O.synth_state_dict(dn=2, seq_len=60)
    model = DanceDecoder(nfeats=151, seq_len=60, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                         cond_feature_dim=438, activation=F.gelu, required_dancer_num=2, compute_dtype="f32")
    model.load_state_dict(sd, strict=True)
    diff = GaussianDiffusion(model.eval(), 60, 151, None, schedule="cosine", n_timestep=100, predict_epsilon=True,
                             clip_denoised=True, loss_type="l2", use_p2=False, cond_drop_prob=0.25, guidance_weight=2,
                             seq_len=60).to(DEV).eval()
    cond = torch.stack([O.synth_cond(0, 60)])
    xT = torch.stack([O.synth_xT(0, 120)])
    value = torch.stack([O.synth_motion(0, 120)]).to(DEV)
    mask = torch.stack([O.synth_inpaint_mask(120)]).to(DEV)
    x = diff.inpaint_loop((1, 120, 151), cond, noise=xT, constraint={"mask": mask, "value": value}, start_point=30,
                          step_noise=dev_noise([0], 120),
                          q_noise=lambda t, shape: torch.stack([O.synth_q_eps(0, t, 120)]))
    e, scale = maxabs(x, ref), float(np.abs(ref).max())
    print(f"C1 inpaint_loop, predict_epsilon=True (30 steps): {e:.2e} (|x| max {scale:.2f})")
    assert e < 1e-3 * max(1.0, scale)


def test_c1_long_inpaint_loop_vs_reference_golden(golden_dir, c1):
    """long_inpaint_loop (model/diffusion.py:560-608): B=2, first half of clip 1 <- second half of clip 0 every step."""
    _, _, diff, _, _ = c1
    ref = gold(golden_dir, "c1_long_inpaint")
    cond = torch.stack([O.synth_cond(c, 60) for c in (0, 1)])
    xT = torch.stack([O.synth_xT(c, 120) for c in (0, 1)])
    x = diff.long_inpaint_loop((2, 120, 151), cond, noise=xT, step_noise=dev_noise([0, 1], 120))
    e = maxabs(x, ref["final"])
    print(f"C1 long_inpaint_loop (100 steps, B=2): {e:.2e}")
    assert e < 1e-3


def test_batch_independence_and_partition_invariance(c2):
    """a clip's sample does not depend on its batch or on the shard it lands in (in-kernel Philox keyed by the
    global clip index): [clip0, clip1] in one batch == clip1 alone with clip_offset=1, bit for bit."""
    _, _, diff, cond, xT = c2
    both = diff.p_sample_loop((2, 450, 151), cond, noise=xT, start_point=6, seed=99, clip_offset=0)
    one = diff.p_sample_loop((1, 450, 151), cond[1:], noise=xT[1:], start_point=6, seed=99, clip_offset=1)
    assert maxabs(both[1:], one) < 2e-5


@pytest.mark.parametrize("compute", ["bf16", "f32"])
def test_full_batch_properties_partition_determinism(compute):
    """BASELINE config 2 at its full batch (16 clips, 3 x 150), 24 DDPM steps across the guidance-clip boundary
    (t = 111..88: both CFG branches, then the conditional one only), in-kernel Philox noise.  Size-independent
    properties, bit for bit: (1) two separate 8-clip calls with clip_offset (what two ranks would compute) give the
    16-clip call's samples, (2) the same seed gives the same samples, a different seed different ones."""
    _, _, diff = build(3, 150, 1000, compute)
    cond = torch.stack([O.synth_cond(c, 150) for c in range(16)])
    xT = torch.stack([O.synth_xT(c, 450) for c in range(16)])
    single = diff.p_sample_loop((16, 450, 151), cond, noise=xT, start_point=112, seed=4242)
    halves = [diff.p_sample_loop((8, 450, 151), cond[lo:lo + 8], noise=xT[lo:lo + 8], start_point=112, seed=4242,
                                 clip_offset=lo) for lo in (0, 8)]
    assert torch.equal(single, torch.cat(halves)), "a clip's sample depends on its shard"
    again = diff.p_sample_loop((16, 450, 151), cond, noise=xT, start_point=112, seed=4242)
    other = diff.p_sample_loop((16, 450, 151), cond, noise=xT, start_point=112, seed=4243)
    assert torch.equal(single, again) and not torch.equal(single, other)
    assert bool(torch.isfinite(single).all()) and float(single.abs().max()) < 20.0


def test_dead_parameters_do_not_change_the_output(c1):
    sd, model, diff, cond, xT = c1
    tt = torch.full((1,), 50, dtype=torch.long, device=DEV)
    a = model(xT.to(DEV), cond.to(DEV), tt)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "traj_Modulation" in n or "traj_embedding" in n or "embeddings_table" in n:
                p.add_(1.0)
    b = model(xT.to(DEV), cond.to(DEV), tt)       # weights re-packed (parameter versions changed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "traj_Modulation" in n or "traj_embedding" in n or "embeddings_table" in n:
                p.sub_(1.0)
    assert torch.equal(a, b)


def test_bf16_mode_against_oracle():
    """throughput mode (bf16 MFMA operands, fp32 accumulate / residual / softmax / LayerNorm): deviation from the
    fp32 oracle is bounded by bf16 operand rounding; stated bound 2.5e-2 max-abs on |x| <= 1 outputs after 20 steps (observed ~1e-2)."""
    sd, model, diff = build(2, 60, 100, compute="bf16")
    cond = torch.stack([O.synth_cond(0, 60)])
    xT = torch.stack([O.synth_xT(0, 120)])
    tt = torch.full((1,), 50, dtype=torch.long)
    ref = O.guided_forward(sd, xT, cond, tt, 2)
    out = model.guided_forward(xT.to(DEV), cond.to(DEV), tt.to(DEV), 2)
    e1 = maxabs(out, ref)
    want = O.p_sample_loop(sd, (1, 120, 151), cond, noise=xT, n_timestep=100, start_point=20,
                           step_noise=O.batch_step_noise([0], 120))
    got = diff.p_sample_loop((1, 120, 151), cond, noise=xT, start_point=20, step_noise=dev_noise([0], 120))
    e2 = maxabs(got, want)
    print(f"bf16 mode vs fp32 oracle: one guided evaluation {e1:.2e}, 20 DDPM steps {e2:.2e}")
    assert e1 < 2.5e-2 and e2 < 2.5e-2


def test_product_path_fails_loudly_off_gpu():
    from tcdiff_amd import _lib
    model = DanceDecoder(nfeats=151, seq_len=60, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8,
                         cond_feature_dim=438, required_dancer_num=2)
    with pytest.raises(_lib.TcdiffError):
        model(torch.zeros(1, 120, 151), torch.zeros(1, 121, 438), torch.zeros(1, dtype=torch.long))


def test_c4_long_sequence_forward_and_steps_vs_oracle():
    """BASELINE config 4 (5 dancers x 300 frames, L = 1500 tokens, 302 memory rows): the streaming attention path
    (K/V do not fit LDS) and the 2560-wide fusion projection, f32 mode against the CPU oracle."""
    dn, S, T = 5, 300, 1000
    sd, model, diff = build(dn, S, T)
    Lq = dn * S
    cond = torch.stack([O.synth_cond(0, S)])
    xT = torch.stack([O.synth_xT(0, Lq)])
    tt = torch.full((1,), 700, dtype=torch.long)
    want = O.guided_forward(sd, xT, cond, tt, 2)
    got = model.guided_forward(xT.to(DEV), cond.to(DEV), tt.to(DEV), 2)
    e = maxabs(got, want)
    print(f"C4 guided forward (L=1500): {e:.2e}")
    assert e < 4e-4
    eps = O.batch_step_noise([0], Lq)
    tab = O.make_tables(T)
    x = xT.clone()
    for i in (999, 998):
        x, _ = O.p_sample(sd, tab, x, cond, i, T, 2, eps(i, x.shape))
    from tcdiff_amd import _lib as L
    chain = []
    diff._run(L.SAMPLER_DDPM, (1, Lq, 151), cond, xT.to(DEV), [999, 998], diff._ddpm_params([999, 998]),
              step_noise=dev_noise([0], Lq), collect=chain)
    e = maxabs(chain[-1], x)
    print(f"C4 two DDPM steps: {e:.2e}")
    assert e < 5e-4


@pytest.mark.parametrize("compute,bound", [("f32", 5e-4), ("bf16x3", 1e-3), ("bf16", 2.5e-2)])
def test_c4_vs_reference_golden(golden_dir, compute, bound):
    """BASELINE config 4 against the REAL reference (tests/golden/make_golden_c4.py): a guided evaluation at t = 500 and
    the first two DDPM steps, 5 dancers x 300 frames."""
    from tcdiff_amd import _lib as L
    dn, S, T = 5, 300, 1000
    ref = gold(golden_dir, "c4_steps")
    _, model, diff = build(dn, S, T, compute=compute)
    Lq = dn * S
    cond = torch.stack([O.synth_cond(0, S)])
    xT = torch.stack([O.synth_xT(0, Lq)])
    tt = torch.full((1,), 500, dtype=torch.long)
    e0 = maxabs(model.guided_forward(xT.to(DEV), cond.to(DEV), tt.to(DEV), 2), ref["guided_w2_t500"])
    chain = []
    diff._run(L.SAMPLER_DDPM, (1, Lq, 151), cond, xT.to(DEV), [999, 998], diff._ddpm_params([999, 998]),
              step_noise=dev_noise([0], Lq), collect=chain)
    e1, e2 = maxabs(chain[-2], ref["after_step_999"]), maxabs(chain[-1], ref["after_step_998"])
    print(f"C4 ({compute}) vs reference golden: guided t=500 {e0:.2e}, after step 999 {e1:.2e}, after step 998 {e2:.2e}")
    assert max(e0, e1, e2) < bound


def test_c4_bf16_runs_and_is_close():
    dn, S = 5, 300
    sd, model, diff = build(dn, S, 1000, compute="bf16")
    cond = torch.stack([O.synth_cond(c, S) for c in (0, 1)])
    xT = torch.stack([O.synth_xT(c, dn * S) for c in (0, 1)])
    tt = torch.full((2,), 500, dtype=torch.long)
    want = O.guided_forward(sd, xT[:1], cond[:1], tt[:1], 2)
    got = model.guided_forward(xT.to(DEV), cond.to(DEV), tt.to(DEV), 2)
    e = maxabs(got[:1], want)
    print(f"C4 bf16 guided forward vs fp32 oracle: {e:.2e}")
    assert e < 2.5e-2 and bool(torch.isfinite(got).all())


# ------------------------------------------------------------------------------------------------------------------
# parity AT the benchmarked configuration (BASELINE config 2: 3 x 150, B = 16, bf16)
# ------------------------------------------------------------------------------------------------------------------
BF16_EVAL_BOUND = 2.5e-2      # one guided evaluation, |x| <= O(1) outputs: bf16 operand rounding through 8 layers
BF16_STEPS_BOUND = 2.5e-2    # sampler state after a few steps / a full DDIM run


@pytest.fixture(scope="module")
def c2_bf16():
    sd, model, diff = build(3, 150, 1000, compute="bf16")
    cond = torch.stack([O.synth_cond(c, 150) for c in (0, 1)])
    xT = torch.stack([O.synth_xT(c, 450) for c in (0, 1)])
    return sd, model, diff, cond, xT


def test_c2_bf16_forward_vs_reference_golden(golden_dir, c2_bf16):
    """the benchmarked arithmetic (bf16 MFMA operands) at the benchmarked shape against the REAL reference's outputs"""
    _, model, _, cond, xT = c2_bf16
    ref = gold(golden_dir, "c2_forward")
    for t in (999, 37):
        tt = torch.full((1,), t, dtype=torch.long, device=DEV)
        e = maxabs(model.guided_forward(xT[:1].to(DEV), cond[:1].to(DEV), tt, 2), ref[f"guided_w2_t{t}"])
        print(f"C2 bf16 guided t={t} vs reference golden: {e:.2e} (bound {BF16_EVAL_BOUND})")
        assert e < BF16_EVAL_BOUND
    out = model(xT.to(DEV), cond.to(DEV), torch.tensor([500, 20], device=DEV), cond_drop_prob=0.0)
    e = maxabs(out, ref["fwd_cond_b2_t500_20"])
    print(f"C2 bf16 forward per-clip timesteps vs reference golden: {e:.2e}")
    assert e < BF16_EVAL_BOUND


def test_c2_bf16_ddpm_steps_and_ddim_vs_reference_golden(golden_dir, c2_bf16):
    _, _, diff, cond, xT = c2_bf16
    from tcdiff_amd import _lib as L
    ref = gold(golden_dir, "c2_ddpm_steps")
    tseq = [999, 998, 997]
    chain = []
    diff._run(L.SAMPLER_DDPM, (1, 450, 151), cond[:1], xT[:1].to(DEV), tseq, diff._ddpm_params(tseq),
              step_noise=dev_noise([0], 450), collect=chain)
    for j, i in enumerate(tseq):
        e = maxabs(chain[j], ref[f"after_step_{i}"])
        print(f"C2 bf16 DDPM step {i} vs reference golden: {e:.2e} (bound {BF16_STEPS_BOUND})")
        assert e < BF16_STEPS_BOUND
    x0 = torch.stack([O.synth_traj(0, 450)])
    x = diff.ddim_sample((1, 450, 151), cond[:1], x_0=x0, init_noise=xT[:1], step_noise=dev_noise([0], 450))
    e = maxabs(x, gold(golden_dir, "c2_ddim")["final"])
    print(f"C2 bf16 ddim_sample (50 steps) vs reference golden: {e:.2e} (bound {BF16_STEPS_BOUND})")
    assert e < BF16_STEPS_BOUND


@pytest.mark.parametrize("compute,bound", [("f32", 5e-4), ("bf16", BF16_STEPS_BOUND)])
def test_c2_full_batch_16_clip0_vs_reference_golden(golden_dir, compute, bound):
    """B = 16 (the benchmarked batch, the single-stream captured step the bench runs) tied to the reference: clip 0 of the
    16-clip batch against the goldens the reference produced for clip 0 alone -- three DDPM steps with injected noise
    and one guided evaluation; clip 1 against the reference's two-clip forward."""
    _, model, diff = build(3, 150, 1000, compute)
    from tcdiff_amd import _lib as L
    ids = list(range(16))
    cond = torch.stack([O.synth_cond(c, 150) for c in ids])
    xT = torch.stack([O.synth_xT(c, 450) for c in ids])
    ref = gold(golden_dir, "c2_ddpm_steps")
    tseq = [999, 998, 997]
    chain = []
    diff._run(L.SAMPLER_DDPM, (16, 450, 151), cond, xT.to(DEV), tseq, diff._ddpm_params(tseq),
              step_noise=dev_noise(ids, 450), collect=chain)
    for j, i in enumerate(tseq):
        e = maxabs(chain[j][:1], ref[f"after_step_{i}"])
        print(f"C2 {compute} B=16 clip 0 after DDPM step {i} vs reference golden: {e:.2e} (bound {bound})")
        assert e < bound
    fwd = gold(golden_dir, "c2_forward")
    tt = torch.full((16,), 999, dtype=torch.long, device=DEV)
    g = model.guided_forward(xT.to(DEV), cond.to(DEV), tt, 2)
    e = maxabs(g[:1], fwd["guided_w2_t999"])
    print(f"C2 {compute} B=16 clip 0 guided evaluation vs reference golden: {e:.2e}")
    assert e < (4e-4 if compute == "f32" else BF16_EVAL_BOUND)
    t16 = torch.full((16,), 700, dtype=torch.long, device=DEV)
    t16[0], t16[1] = 500, 20
    out = model(xT.to(DEV), cond.to(DEV), t16, cond_drop_prob=0.0)
    e = maxabs(out[:2], fwd["fwd_cond_b2_t500_20"])
    print(f"C2 {compute} B=16 clips 0,1 conditional forward vs reference golden: {e:.2e}")
    assert e < (2e-4 if compute == "f32" else BF16_EVAL_BOUND)


BF16_DRIFT_BOUND = 3e-2      # observed on MI355X: max-abs 8.3e-3, mean-abs 1.7e-3


def test_c2_bf16_vs_f32_drift_over_the_full_1000_steps():
    """the metric is defined on 1000 DDPM steps: run the benchmarked bf16 mode and the f32 parity mode through ALL of them
    at the benchmarked shape (B = 2, same Philox seed, in-kernel noise) and bound how far bf16 operand rounding has
    drifted at the end.  (The loop is contractive -- SURVEY.md section 4 -- so the drift does not compound.)"""
    cond = torch.stack([O.synth_cond(c, 150) for c in (0, 1)])
    xT = torch.stack([O.synth_xT(c, 450) for c in (0, 1)])
    outs = {}
    for compute in ("f32", "bf16"):
        _, _, diff = build(3, 150, 1000, compute)
        outs[compute] = diff.p_sample_loop((2, 450, 151), cond, noise=xT, seed=777)
    d = (outs["bf16"] - outs["f32"]).abs()
    e, mean = float(d.max()), float(d.mean())
    print(f"C2 bf16 vs f32 mode after 1000 DDPM steps (B=2, seed 777): max-abs {e:.3e}, mean-abs {mean:.3e} "
          f"(bound {BF16_DRIFT_BOUND})")
    assert bool(torch.isfinite(outs["bf16"]).all())
    assert e < BF16_DRIFT_BOUND and mean < BF16_DRIFT_BOUND / 10


def test_first_call_crossing_the_guidance_boundary_equals_the_replayed_call():
    """On the FIRST call of a process every step graph key is met eagerly once, then captured, then replayed -- and at
    t < 0.1 T the key changes (single-branch steps) after hundreds of replays of the two-branch graph.  The first call on a
    fresh model (eager warm-up + capture inside the loop) must give, bit for bit, what the all-replayed second call and the
    never-captured schedule give."""
    cond = torch.stack([O.synth_cond(c, 150) for c in range(4)])
    xT = torch.stack([O.synth_xT(c, 450) for c in range(4)])
    _, _, fresh = build(3, 150, 1000, "bf16")
    first = fresh.p_sample_loop((4, 450, 151), cond, noise=xT, start_point=400, seed=31)     # first call: nothing captured
    again = fresh.p_sample_loop((4, 450, 151), cond, noise=xT, start_point=400, seed=31)     # every step replayed
    assert torch.equal(first, again)
    _, _, ref = build(3, 150, 1000, "bf16")
    eager = ref.p_sample_loop((4, 450, 151), cond, noise=xT, start_point=400, seed=31, use_graph=False)
    assert torch.equal(first, eager)


@pytest.mark.parametrize("steps", ["7", "20"])
def test_several_steps_per_captured_graph_equal_one_step_per_graph(monkeypatch, steps):
    """Round 6: runs of TCDIFF_GRAPH_STEPS consecutive steps of one kind replay ONE graph (every per-step quantity is in device
    memory); run tails and the guidance boundary (t < 0.1 T: single-branch steps) fall back to single steps.  Bit-identical to
    one step per graph, on the first call (eager warm-up, capture) and on the replayed one."""
    cond = torch.stack([O.synth_cond(c, 150) for c in range(3)])
    xT = torch.stack([O.synth_xT(c, 450) for c in range(3)])
    outs = {}
    for n in ("1", steps):
        monkeypatch.setenv("TCDIFF_GRAPH_STEPS", n)
        _, _, d = build(3, 150, 1000, "bf16")
        first = d.p_sample_loop((3, 450, 151), cond, noise=xT, start_point=263, seed=5)       # 163 two-branch + 100 single-branch steps
        again = d.p_sample_loop((3, 450, 151), cond, noise=xT, start_point=263, seed=5)
        assert torch.equal(first, again)
        outs[n] = first
    assert torch.equal(outs["1"], outs[steps])


def test_weights_written_by_fused_ema_and_adan_are_seen_by_the_next_forward():
    """The fused EMA / Adan kernels write parameters through raw pointers; the packed model-dtype weight copies,
    conditioning caches and captured graphs are keyed by Parameter._version, which those updates must bump
    (model/diffusion.py:61-76 feeds master_model, which the samplers then run: TCDiff.py:243-245,277-303)."""
    from tcdiff_amd.adan import Adan
    from tcdiff_amd.diffusion import EMA
    _, ma, _ = build(2, 60, 100, compute="bf16")
    _, cur, _ = build(2, 60, 100, compute="bf16")
    ma, cur = ma.to(DEV), cur.to(DEV)
    cond = torch.stack([O.synth_cond(0, 60)]).to(DEV)
    x = torch.stack([O.synth_xT(0, 120)]).to(DEV)
    t = torch.full((1,), 50, dtype=torch.long, device=DEV)
    y0 = ma(x, cond, t).clone()
    with torch.no_grad():
        for p in cur.parameters():
            p.mul_(1.05)
    y_cur = cur(x, cond, t).clone()
    assert maxabs(y_cur, y0) > 1e-3
    v0 = [p._version for p in ma.parameters()]
    EMA(0.0).update_model_average(ma, cur)                    # beta = 0: ma <- cur exactly
    assert all(p._version > v for p, v in zip(ma.parameters(), v0))
    assert torch.equal(ma(x, cond, t), y_cur)                 # not the stale packed weights
    for p in ma.parameters():
        p.grad = torch.full_like(p, 1e-3)
    opt = Adan(ma.parameters(), lr=1e-2, weight_decay=0.02)
    opt.step(); opt.step()                                    # second step moves the weights (model/adan.py:71)
    assert maxabs(ma(x, cond, t), y_cur) > 1e-4


@pytest.mark.parametrize("compute,bound", [("f32", 1e-3), ("bf16x3", 1e-3), ("bf16", 1.5e-2)])
def test_c2_full_1000_step_loop_vs_reference_golden(golden_dir, compute, bound):
    """The north-star claim at the benchmark's own length: one clip of 3 dancers x 150 frames through ALL 1000 DDPM steps
    of the REAL reference's p_sample_loop with injected noise (tests/golden/make_golden_c2_full.py), here as clip 0 of a
    two-clip batch.  f32 mode AND the split-bf16 mode ("bf16x3": fp32 storage, three bf16 MFMAs per product): max-abs <= 1e-3
    (the north-star's tolerance) at every checkpoint and at the end; bf16 (the benchmarked mode): its own stated bound, with
    the observed deviation from the reference printed."""
    ref = gold(golden_dir, "c2_p_sample_loop_full")
    _, _, diff = build(3, 150, 1000, compute=compute)
    cond = torch.stack([O.synth_cond(c, 150) for c in (0, 1)])
    xT = torch.stack([O.synth_xT(c, 450) for c in (0, 1)])
    x, chain = diff.p_sample_loop((2, 450, 151), cond, noise=xT, step_noise=dev_noise([0, 1], 450), return_diffusion=True)
    errs = {}
    for k in ref.files:
        if k.startswith("after_step_"):
            i = int(k.split("_")[-1])
            errs[k] = maxabs(chain[1000 - i][:1], ref[k])
    errs["final"] = maxabs(x[:1], ref["final"])
    print(f"C2 full loop ({compute}) max-abs vs reference after steps:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) < bound


def test_bf16_error_attribution_by_switch(golden_dir):
    """where does the bf16 mode's distance to the reference come from?  One guided evaluation of clip 0 at the
    benchmarked shape against the REAL reference's output (tests/golden/c2_forward.npz), for the default bf16 path and with
    one ingredient switched off at a time: the two weight folds (input_projection . fusion linear 1, linear3 . final_layer:
    products taken in fp32 and rounded to bf16 ONCE), the front chain launch, the fused layer chain (A + B launches with the
    op-by-op attention kernel between them) and all chains (op-by-op kernels: every intermediate rounded to bf16 in HBM).
    Prints the table (profiles/r04_parity_at_benchmarked_config.log) and holds every variant to the stated bound; the f32
    mode is the reference point that carries the <= 1e-3 claim."""
    import importlib
    ref = gold(golden_dir, "c2_forward")
    cond = torch.stack([O.synth_cond(0, 150)]).to(DEV)
    xT = torch.stack([O.synth_xT(0, 450)]).to(DEV)
    variants = [("bf16 default", {}), ("bf16, no input fold", {"TCDIFF_FOLD_IN": "0"}), ("bf16, no output fold", {"TCDIFF_FOLD_OUT": "0"}),
                ("bf16, no front chain", {"TCDIFF_FRONT": "0"}), ("bf16, chain A + B", {"TCDIFF_CHAIN": "1"}),
                ("bf16, op-by-op", {"TCDIFF_CHAIN": "0"}), ("f32 parity mode", None)]
    keys = ("TCDIFF_FOLD_IN", "TCDIFF_FOLD_OUT", "TCDIFF_FRONT", "TCDIFF_CHAIN")
    saved = {k: os.environ.get(k) for k in keys}
    rows = []
    try:
        for name, env in variants:
            for k in keys:
                os.environ.pop(k, None)
            os.environ.update(env or {})
            _, model, _ = build(3, 150, 1000, compute="f32" if env is None else "bf16")
            errs = []
            for t in (999, 37):
                tt = torch.full((1,), t, dtype=torch.long, device=DEV)
                y = model.guided_forward(xT, cond, tt, 2)
                d = (y.detach().cpu().double().numpy() - ref[f"guided_w2_t{t}"])
                errs.append((float(np.abs(d).max()), float(np.abs(d).mean())))
            rows.append((name, errs))
            del model
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    print("\\nguided evaluation vs the real reference (clip 0, 3 x 150, w = 2): max-abs / mean-abs at t = 999 | t = 37")
    for name, errs in rows:
        print(f"  {name:24s} {errs[0][0]:.3e} / {errs[0][1]:.3e} | {errs[1][0]:.3e} / {errs[1][1]:.3e}")
    worst = {name: max(e[0] for e in errs) for name, errs in rows}
    assert worst["f32 parity mode"] < 5e-4
    for name, e in worst.items():
        if name != "f32 parity mode":
            assert e < BF16_EVAL_BOUND, (name, e)
    # no single ingredient of the default path costs more than half of its distance to the reference: the error is the bf16
    # operand rounding of eight layers, not one of the folds
    base = worst["bf16 default"]
    assert all(base - e < 0.5 * base for n, e in worst.items() if n.startswith("bf16,")), worst


# ---- the split-bf16 mode (compute_dtype = "bf16x3": fp32 storage, every GEMM / attention product as three bf16 MFMAs on (hi, lo)
# splits, csrc/common.h MmaBF16x3) against the REAL reference's goldens at the north-star's tolerance, 1e-3 -------------------
@pytest.fixture(scope="module")
def c2_x3():
    sd, model, diff = build(3, 150, 1000, "bf16x3")
    cond = torch.stack([O.synth_cond(c, 150) for c in (0, 1)])
    xT = torch.stack([O.synth_xT(c, 450) for c in (0, 1)])
    return sd, model, diff, cond, xT


def test_bf16x3_c1_forward_and_loop_vs_reference_golden(golden_dir):
    _, model, diff = build(2, 60, 100, "bf16x3")
    cond = torch.stack([O.synth_cond(0, 60)])
    xT = torch.stack([O.synth_xT(0, 120)])
    ref = gold(golden_dir, "c1_forward")
    for t in (99, 3):
        tt = torch.full((1,), t, dtype=torch.long, device=DEV)
        e_c = maxabs(model(xT.to(DEV), cond.to(DEV), tt, cond_drop_prob=0.0), ref[f"fwd_cond_t{t}"])
        e_g = maxabs(model.guided_forward(xT.to(DEV), cond.to(DEV), tt, 2), ref[f"guided_w2_t{t}"])
        print(f"bf16x3 C1 forward t={t}: cond {e_c:.2e} guided {e_g:.2e}")
        assert e_c < 1e-3 and e_g < 1e-3
    ref = gold(golden_dir, "c1_p_sample_loop")
    x, chain = diff.p_sample_loop((1, 120, 151), cond, noise=xT, step_noise=dev_noise([0], 120), return_diffusion=True)
    errs = {k: maxabs(chain[i], ref[k]) for k, i in (("after_step_99", 1), ("after_step_50", 50), ("after_step_10", 90),
                                                    ("after_step_1", 99))}
    errs["final"] = maxabs(x, ref["final"])
    print("bf16x3 C1 p_sample_loop max-abs vs reference:", {k: f"{v:.2e}" for k, v in errs.items()})
    assert max(errs.values()) < 1e-3


def test_bf16x3_c2_forward_steps_and_ddim_vs_reference_golden(golden_dir, c2_x3):
    _, model, diff, cond, xT = c2_x3
    ref = gold(golden_dir, "c2_forward")
    for t in (999, 37):
        tt = torch.full((1,), t, dtype=torch.long, device=DEV)
        e = maxabs(model.guided_forward(xT[:1].to(DEV), cond[:1].to(DEV), tt, 2), ref[f"guided_w2_t{t}"])
        print(f"bf16x3 C2 guided t={t}: {e:.2e}")
        assert e < 1e-3
    out = model(xT.to(DEV), cond.to(DEV), torch.tensor([500, 20], device=DEV), cond_drop_prob=0.0)
    e = maxabs(out, ref["fwd_cond_b2_t500_20"])
    print(f"bf16x3 C2 forward per-clip timesteps: {e:.2e}")
    assert e < 1e-3
    ref = gold(golden_dir, "c2_ddpm_steps")
    from tcdiff_amd import _lib as L
    tseq = [999, 998, 997]
    chain = []
    diff._run(L.SAMPLER_DDPM, (1, 450, 151), cond[:1], xT[:1].to(DEV), tseq, diff._ddpm_params(tseq),
              step_noise=dev_noise([0], 450), collect=chain)
    for j, i in enumerate(tseq):
        e = maxabs(chain[j], ref[f"after_step_{i}"])
        print(f"bf16x3 C2 DDPM step {i}: {e:.2e}")
        assert e < 1e-3
    x0 = torch.stack([O.synth_traj(0, 450)])
    x = diff.ddim_sample((1, 450, 151), cond[:1], x_0=x0, init_noise=xT[:1], step_noise=dev_noise([0], 450))
    e = maxabs(x, gold(golden_dir, "c2_ddim")["final"])
    print(f"bf16x3 C2 ddim_sample (50 steps, trajectory in-painting): {e:.2e}")
    assert e < 1e-3
