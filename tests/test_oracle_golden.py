"""Pin the CPU oracle (oracle/tcdiff_oracle.py) to golden vectors produced by the REAL reference
(tests/golden/make_golden.py, run in the build container against /root/reference).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import tcdiff_oracle as O

torch.set_num_threads(max(1, min(8, os.cpu_count() or 1)))


def g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def maxabs(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))))


@pytest.mark.parametrize("T", [100, 1000])
def test_schedule_tables_bit_exact(golden_dir, T):
    ref = g(golden_dir, f"tables_T{T}")
    tab = O.make_tables(T, "cosine")
    for k in ref.files:
        assert np.array_equal(ref[k], tab[k].numpy()), k


def test_state_dict_keys_and_shapes(golden_dir):
    ref = g(golden_dir, "state_dict_keys_dn3")
    shapes = O.reference_param_shapes(dn=3, seq_len=150)
    assert sorted(shapes) == list(ref["keys"])
    for k, s in zip(ref["keys"], ref["shapes"]):
        assert str(tuple(shapes[str(k)])) == str(s), k
    n_params = sum(int(np.prod(shapes[str(k)])) for k in ref["param_order"])
    assert n_params == 61428181


@pytest.fixture(scope="module")
def c1():
    sd = O.synth_state_dict(dn=2, seq_len=60)
    cond = torch.stack([O.synth_cond(0, 60)])
    xT = torch.stack([O.synth_xT(0, 120)])
    return sd, cond, xT


def test_c1_blocks(golden_dir, c1):
    sd, _, _ = c1
    ref = g(golden_dir, "c1_blocks")
    gen = torch.Generator().manual_seed(77)
    xb = torch.randn(1, 120, 512, generator=gen)
    mem = torch.randn(1, 62, 512, generator=gen)
    tb = torch.randn(1, 512, generator=gen)
    xe = torch.randn(1, 60, 512, generator=gen)
    fr = sd["rotary.freqs"]
    p = "seqTransDecoder.stack.0"
    assert maxabs(O.rotary(xb, fr), ref["rotary_x"]) < 1e-5
    assert maxabs(O.sinusoidal_emb(torch.tensor([0, 1, 37, 99, 999]), 512), ref["sinusoidal"]) < 1e-6
    assert maxabs(O.decoder_layer(xb, mem, tb, sd, p, fr, 8), ref["decoder_layer0"]) < 2e-5
    assert maxabs(O.encoder_layer(xe, sd, "cond_encoder.0", fr, 8), ref["encoder_layer0"]) < 2e-5
    assert maxabs(O.sbi_msa(xb, xb, xb, sd, p + ".self_attn", 8), ref["sbi_self"]) < 2e-5
    sc, sh = O.film(tb, sd, p + ".film1")
    assert maxabs(sc, ref["film1_scale"]) < 1e-5 and maxabs(sh, ref["film1_shift"]) < 1e-5


def test_c1_forward(golden_dir, c1):
    sd, cond, xT = c1
    ref = g(golden_dir, "c1_forward")
    for t in (99, 3):
        tt = torch.full((1,), t, dtype=torch.long)
        assert maxabs(O.decoder_forward(sd, xT, cond, tt, 0.0), ref[f"fwd_cond_t{t}"]) < 2e-5
        assert maxabs(O.decoder_forward(sd, xT, cond, tt, 1.0), ref[f"fwd_unc_t{t}"]) < 2e-5
        assert maxabs(O.guided_forward(sd, xT, cond, tt, 2), ref[f"guided_w2_t{t}"]) < 5e-5


def test_c1_p_sample_loop(golden_dir, c1):
    """BASELINE config 1: 1 clip, 2 dancers x 60 frames, 100 DDPM steps, injected noise."""
    sd, cond, xT = c1
    ref = g(golden_dir, "c1_p_sample_loop")
    x, chain = O.p_sample_loop(sd, (1, 120, 151), cond, noise=xT, n_timestep=100,
                               step_noise=O.batch_step_noise([0], 120), return_diffusion=True)
    assert maxabs(chain[1], ref["after_step_99"]) < 1e-4
    assert maxabs(chain[50], ref["after_step_50"]) < 1e-4
    assert maxabs(chain[90], ref["after_step_10"]) < 1e-4
    assert maxabs(chain[99], ref["after_step_1"]) < 1e-4
    assert maxabs(x, ref["final"]) < 1e-4


@pytest.fixture(scope="module")
def c2():
    sd = O.synth_state_dict(dn=3, seq_len=150)
    cond = torch.stack([O.synth_cond(c, 150) for c in (0, 1)])
    xT = torch.stack([O.synth_xT(c, 450) for c in (0, 1)])
    return sd, cond, xT


def test_c2_full_loop_first_100_steps(golden_dir, c2):
    """The benchmark-length golden (tests/golden/make_golden_c2_full.py: the real reference's p_sample_loop, 1 clip,
    3 x 150, 1000 steps): the oracle is walked through the first 100 steps here (the GPU suite runs all 1000)."""
    sd, cond, xT = c2
    ref = g(golden_dir, "c2_p_sample_loop_full")
    tab = O.make_tables(1000, "cosine")
    eps = O.batch_step_noise([0], 450)
    x = xT[:1].clone()
    for i in reversed(range(900, 1000)):
        x, _ = O.p_sample(sd, tab, x, cond[:1], i, 1000, 2, eps(i, x.shape))
        if i == 999:
            assert maxabs(x, ref["after_step_999"]) < 1e-4
    assert maxabs(x, ref["after_step_900"]) < 1e-4


def test_c2_forward(golden_dir, c2):
    sd, cond, xT = c2
    ref = g(golden_dir, "c2_forward")
    for t in (999, 37):
        tt = torch.full((1,), t, dtype=torch.long)
        assert maxabs(O.guided_forward(sd, xT[:1], cond[:1], tt, 2), ref[f"guided_w2_t{t}"]) < 5e-5
    out = O.decoder_forward(sd, xT, cond, torch.tensor([500, 20]), 0.0)
    assert maxabs(out, ref["fwd_cond_b2_t500_20"]) < 2e-5


def test_c2_ddpm_steps(golden_dir, c2):
    sd, cond, xT = c2
    ref = g(golden_dir, "c2_ddpm_steps")
    tab = O.make_tables(1000)
    eps = O.batch_step_noise([0], 450)
    for start in (1000, 3):
        x = xT[:1].clone()
        for i in reversed(range(start - 3, start)):
            x, _ = O.p_sample(sd, tab, x, cond[:1], i, 1000, 2, eps(i, x.shape))
            assert maxabs(x, ref[f"after_step_{i}"]) < 1e-4, i


def test_c2_ddim_with_trajectory(golden_dir, c2):
    sd, cond, xT = c2
    ref = g(golden_dir, "c2_ddim")
    x0 = torch.stack([O.synth_traj(0, 450)])
    x = O.ddim_sample(sd, (1, 450, 151), cond[:1], x_0=x0, init_noise=xT[:1],
                      step_noise=O.batch_step_noise([0], 450))
    assert maxabs(x, ref["final"]) < 1e-4


def test_c2_long_ddim(golden_dir, c2):
    sd, cond, xT = c2
    ref = g(golden_dir, "c2_long_ddim")
    x0 = torch.stack([O.synth_traj(c, 450) for c in (0, 1)]).reshape(2, 150, 3, 3)
    x = O.long_ddim_sample(sd, (2, 450, 151), cond, x0, seq_len=150, init_noise=xT,
                           step_noise=O.batch_step_noise([0, 1], 450))
    assert maxabs(x, ref["final"]) < 1e-4


def test_c2_footwork_ddim(golden_dir, c2):
    """ddim_sample_Footwork (model/diffusion.py:289-383), golden from tests/golden/make_golden_inpaint.py."""
    sd, cond, xT = c2
    ref = g(golden_dir, "c2_footwork")
    x0 = torch.stack([O.synth_motion(0, 450)])
    x = O.ddim_sample_footwork(sd, (1, 450, 151), cond[:1], x_0=x0, init_noise=xT[:1],
                               step_noise=O.batch_step_noise([0], 450))
    assert maxabs(x, ref["final"]) < 1e-4


def test_c1_inpaint_loop(golden_dir, c1):
    """inpaint_loop (model/diffusion.py:519-557) with the q_sample draws injected."""
    sd, cond, xT = c1
    ref = g(golden_dir, "c1_inpaint")
    value = torch.stack([O.synth_motion(0, 120)])
    mask = torch.stack([O.synth_inpaint_mask(120)])
    x = O.inpaint_loop(sd, (1, 120, 151), cond, xT, mask, value, n_timestep=100,
                       step_noise=O.batch_step_noise([0], 120),
                       q_noise=lambda i, shape: torch.stack([O.synth_q_eps(0, i, 120)]))
    assert maxabs(x, ref["final"]) < 1e-4


def test_c1_long_inpaint_loop(golden_dir):
    """long_inpaint_loop (model/diffusion.py:560-608), two half-overlapping windows."""
    sd = O.synth_state_dict(dn=2, seq_len=60)
    cond = torch.stack([O.synth_cond(c, 60) for c in (0, 1)])
    xT = torch.stack([O.synth_xT(c, 120) for c in (0, 1)])
    ref = g(golden_dir, "c1_long_inpaint")
    x = O.long_inpaint_loop(sd, (2, 120, 151), cond, xT, n_timestep=100, step_noise=O.batch_step_noise([0, 1], 120))
    assert maxabs(x, ref["final"]) < 1e-4



def test_c4_guided_forward_and_steps(golden_dir):
    """BASELINE config 4 (5 x 300, L = 1500): oracle against the real reference (tests/golden/make_golden_c4.py)."""
    dn, S, T = 5, 300, 1000
    ref = g(golden_dir, "c4_steps")
    sd = O.synth_state_dict(dn=dn, seq_len=S)
    cond = torch.stack([O.synth_cond(0, S)])
    xT = torch.stack([O.synth_xT(0, dn * S)])
    tt = torch.full((1,), 500, dtype=torch.long)
    assert maxabs(O.guided_forward(sd, xT, cond, tt, 2), ref["guided_w2_t500"]) < 5e-5
    tab = O.make_tables(T)
    eps = O.batch_step_noise([0], dn * S)
    x = xT.clone()
    for i in (999, 998):
        x, _ = O.p_sample(sd, tab, x, cond, i, T, 2, eps(i, x.shape))
        assert maxabs(x, ref[f"after_step_{i}"]) < 1e-4


def test_use_rotary_false_option(golden_dir):
    """DanceDecoder(use_rotary=False) (model/model.py:441-448,564,580; outside the production configuration): the oracle's
    restatement -- no rotation, PositionalEncoding rows added to the motion and the music tokens -- against the REAL reference built
    with that option (tests/golden/make_golden_abs_pos.py): state_dict surface, guided evaluation, conditional forward."""
    ref = g(golden_dir, "c1_abs_pos")
    shapes = O.reference_param_shapes(dn=2, seq_len=60, use_rotary=False)
    assert sorted(shapes) == [str(k) for k in ref["state_dict_keys"]]
    assert "abs_pos_encoding.pe" in shapes and not any(k.endswith("rotary.freqs") for k in shapes)
    sd = O.synth_state_dict(dn=2, seq_len=60, use_rotary=False)
    cond = torch.stack([O.synth_cond(0, 60)])
    xT = torch.stack([O.synth_xT(0, 120)])
    with torch.no_grad():
        e1 = maxabs(O.guided_forward(sd, xT, cond, torch.tensor([50]), 2), ref["guided_w2_t50"])
        e2 = maxabs(O.decoder_forward(sd, xT, cond, torch.tensor([3]), cond_drop_prob=0.0), ref["fwd_cond_t3"])
    assert e1 < 2e-5 and e2 < 2e-5, (e1, e2)
