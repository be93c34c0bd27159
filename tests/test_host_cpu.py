"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol the header declares, argument
validation happens before any launch, the host classes mirror the reference's parameter surface and schedule
tables, and the product path refuses to run off-GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tcdiff_amd import _lib as L
from tcdiff_amd.diffusion import GaussianDiffusion
from tcdiff_amd.model import DanceDecoder

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from tcdiff_amd import build
    build.build(verbose=False)
    return L.load()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "tcdiff_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(tcdiff_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 16
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/tcdiff_hip.h but not exported"
    assert sorted(L.EXPORTS) == declared
    assert b"gfx950" in lib.tcdiff_version()


def test_argument_validation_without_gpu(lib):
    e = L.TileEpi()
    assert lib.tcdiff_gemm_tile(L.DT_BF16, None, None, 0, None, 1, 1, 64, 64, 64, 0, ctypes.byref(e), None) == -1
    r = L.RowEpi()
    assert lib.tcdiff_gemm_rowln(L.DT_F32, None, None, 1, 32, 32, 32, 0, ctypes.byref(r), None) == -1
    assert lib.tcdiff_attention(L.DT_F32, None, None, None, None, 1, 8, 1, 1, 128, 128, 512, 0, 0, None) == -1
    assert lib.tcdiff_ln_rot(L.DT_F32, None, 1, None, None, 1e-5, None, None, None, None, 0, 0, None) == -1
    assert lib.tcdiff_step_end(None, None) == -1
    assert lib.tcdiff_gemm_rows(None, None, 0, None, 64, 512, 512, 512, ctypes.byref(e), 0, None) == -1
    buf = ctypes.create_string_buffer(64)                 # (a non-NULL 16-byte aligned address; nothing is launched)
    al = ctypes.c_void_p((ctypes.addressof(buf) + 15) & ~15)
    assert lib.tcdiff_gemm_rows(al, None, 0, al, 64, 512, 768, 768, ctypes.byref(e), 0, None) == -4     # K = 768: unsupported
    assert lib.tcdiff_gemm_rows(al, None, 0, al, 64, 500, 512, 512, ctypes.byref(e), 0, None) == -4     # N % 512
    assert lib.tcdiff_pack_row_streams(None, 1, 2048, None) == -1
    with pytest.raises(L.TcdiffError):
        L.check(-2, "x")


def test_ctypes_structures_match_the_header(tmp_path):
    """The argument structures cross the C ABI by value / by pointer: the ctypes mirrors in _lib.py must have the header's sizes and
    (for the largest one) field offsets -- checked with the C compiler itself."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("needs gcc and the HIP headers")
    pairs = [("tcdiff_tile_epi", L.TileEpi), ("tcdiff_row_epi", L.RowEpi), ("tcdiff_chain_args", L.ChainArgs),
             ("tcdiff_step_prologue_args", L.StepPrologueArgs), ("tcdiff_row_args", L.RowArgs), ("tcdiff_ct_desc", L.CtDesc),
             ("tcdiff_ws_desc", L.WsDesc), ("tcdiff_tn_problem", L.TnProblem), ("tcdiff_adan_scalars", L.AdanScalars)]
    chain_fields = [f[0] for f in L.ChainArgs._fields_]
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "tcdiff_hip.h"', "int main(void) {"]
    src += [f'  printf("%zu\\n", sizeof({c}));' for c, _ in pairs]
    src += [f'  printf("%zu\\n", offsetof(tcdiff_chain_args, {f}));' for f in chain_fields]
    src += ["  return 0;", "}"]
    cfile, exe = tmp_path / "abi.c", tmp_path / "abi"
    cfile.write_text("\n".join(src))
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__", str(cfile),
                    "-o", str(exe)], check=True, capture_output=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    for (cname, cls), size in zip(pairs, out):
        assert ctypes.sizeof(cls) == size, f"{cname}: header {size} bytes, ctypes mirror {ctypes.sizeof(cls)}"
    for f, off in zip(chain_fields, out[len(pairs):]):
        assert getattr(L.ChainArgs, f).offset == off, f"tcdiff_chain_args.{f}: header offset {off}, ctypes {getattr(L.ChainArgs, f).offset}"


def test_chain_launcher_validates_the_self_attention_arguments_without_gpu(lib):
    """tcdiff_chain refuses inconsistent seq_blocks / sa_* / *f_out combinations before it touches the device (-1); a consistent set
    passes validation and fails only at the device query (-4 here: no GPU)."""
    P = 0x1000

    def rc(**kw):
        a = L.ChainArgs()
        base = dict(mode=L.CHAIN_FULL, n_stages=176, M=450, L=150, H=8, Lp=256, A=P, wstream=P, film=P, xres=P, xout=P, n2_g=P,
                    n2_b=P, rope=P, b1=P, film3=P, n4_g=P, n4_b=P, b3=P, nn_g=P, nn_b=P, film_ld=6144, filmb=P, n3_g=P, n3_b=P, kf=P,
                    vf=P, n_shared=1, nkt=2, Lk=62, rope_rows=150, dn=1, mt=4)
        base.update(kw)
        for k, v in base.items():
            setattr(a, k, v)
        return lib.tcdiff_chain(ctypes.byref(a), None)

    frag = dict(qf_out=P, kf_out=P, vf_out=P, out_nkt=5)
    sa = dict(sa_q=P, sa_kf=P, sa_vf=P, sa_nkt=5)
    if torch.cuda.is_available():
        pytest.skip("valid argument sets would launch on fake pointers")
    assert rc(q_out=P, k_out=P, v_out=P) == -4                                  # the plain fused launch
    assert rc(q_out=P, k_out=P, v_out=P, seq_blocks=1) == -4
    assert rc(seq_blocks=1, **frag) == -4 and rc(seq_blocks=1, **frag, **sa) == -4
    assert rc(mode=L.CHAIN_FULL_LAST, n_stages=128, h_out=P, seq_blocks=1, **sa) == -4
    assert rc(**frag) == -1                                                     # fragment outputs need sequence-cut blocks
    assert rc(seq_blocks=1, qf_out=P, out_nkt=5, q_out=P, k_out=P, v_out=P) == -1      # all three images or none
    assert rc(seq_blocks=1, qf_out=P, kf_out=P, vf_out=P, out_nkt=4) == -1      # 150 keys need 5 tiles
    assert rc(q_out=P, k_out=P, v_out=P, seq_blocks=1, M=444) == -1             # whole sequences only
    assert rc(q_out=P, k_out=P, v_out=P, **sa) == -1                            # the in-kernel attention needs sequence-cut blocks
    assert rc(seq_blocks=1, **frag, sa_q=P, sa_kf=P, sa_vf=P, sa_nkt=4) == -1
    assert rc(seq_blocks=1, nw=4, **frag) == -4                                 # (unsupported in the four-wave form -- also -4)
    assert rc(seq_blocks=1, **frag, sa_q=P + 8, sa_kf=P, sa_vf=P, sa_nkt=5) == -2      # alignment


def test_stream_table_validation_without_gpu(lib):
    """tcdiff_pack_row_streams' descriptor table is checked on the host before anything is launched"""
    from tcdiff_amd import kernels as K
    w = torch.zeros(512, 1024)
    dst = torch.zeros(8, 32, 2048, dtype=torch.bfloat16)
    tab, n, mx = K.ws_table([dict(src=w, sn=1024, sk=1, N=512, K=1024, dst=dst)], "cpu")
    assert n == 1 and mx == 512 * 1024 and tab.numel() == ctypes.sizeof(L.WsDesc)
    with pytest.raises(L.TcdiffError):                     # N not a multiple of 512
        K.ws_table([dict(src=w[:500], sn=1024, sk=1, N=500, K=1024, dst=dst)], "cpu")
    with pytest.raises(L.TcdiffError):                     # a piece beyond the stream's k-steps
        K.ws_table([dict(src=w, sn=1024, sk=1, N=512, K=1024, dst=dst, kst_dst=32, ks0=16)], "cpu")
    with pytest.raises(L.TcdiffError):                     # destination of the wrong size
        K.ws_table([dict(src=w, sn=1024, sk=1, N=512, K=1024, dst=dst[:, :16])], "cpu")
    with pytest.raises(L.TcdiffError):                     # fp32 sources only
        K.ws_table([dict(src=w.double(), sn=1024, sk=1, N=512, K=1024, dst=dst)], "cpu")


@pytest.fixture(scope="module")
def small():
    model = DanceDecoder(nfeats=151, seq_len=150, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                         cond_feature_dim=438, activation=F.gelu, required_dancer_num=3)
    diff = GaussianDiffusion(model, 150, 151, None, schedule="cosine", n_timestep=1000, predict_epsilon=False,
                             loss_type="l2", use_p2=False, cond_drop_prob=0.25, guidance_weight=2, seq_len=150)
    return model, diff


def test_state_dict_surface_matches_reference(golden_dir, small):
    model, diff = small
    ref = np.load(os.path.join(golden_dir, "state_dict_keys_dn3.npz"))
    sd = model.state_dict()
    assert sorted(sd) == list(ref["keys"])
    for k, s in zip(ref["keys"], ref["shapes"]):
        assert str(tuple(sd[str(k)].shape)) == str(s), k
    assert [n for n, _ in model.named_parameters()] == list(ref["param_order"])   # Adan / EMA zip order
    assert sum(p.numel() for p in model.parameters()) == 61428181
    # the deep-copied EMA model exposes the same surface (TCDiff.py:267)
    assert sorted(diff.master_model.state_dict()) == sorted(sd)
    # module.-prefixed checkpoints load with strict=False like TCDiff.py:113-120
    missing, unexpected = model.load_state_dict({("module." + k): v for k, v in sd.items()}, strict=False)
    assert len(unexpected) == len(sd)


def test_schedule_tables_bit_exact(golden_dir, small):
    _, diff = small
    ref = np.load(os.path.join(golden_dir, "tables_T1000.npz"))
    for k in ref.files:
        assert np.array_equal(ref[k], getattr(diff, k).numpy()), k


def test_guidance_clipping_and_ddim_schedule(small):
    _, diff = small
    assert diff._guidance_weight_at(999) == 2 and diff._guidance_weight_at(100) == 2
    assert diff._guidance_weight_at(99) == 1 and diff._guidance_weight_at(0) == 1
    pairs = diff._ddim_pairs()
    assert len(pairs) == 50 and pairs[0] == (999, 979) and pairs[-1][1] == -1
    p = diff._ddim_params(pairs, [2] * 50)
    assert p[-1, 6] == 1 and float(p[:-1, 6].abs().max()) == 0
    from oracle import tcdiff_oracle as O
    assert pairs == O.ddim_time_pairs(1000)
    ddpm = diff._ddpm_params([999, 5, 0])
    assert ddpm[2, 3] == 0 and ddpm[0, 0] == 2 and ddpm[1, 0] == 1
    tab = O.make_tables(1000)
    assert float(ddpm[0, 1]) == float(tab["posterior_mean_coef1"][999])


def test_use_rotary_false_surface(golden_dir):
    """DanceDecoder(use_rotary=False): the reference's state_dict surface for that option (no rotary.freqs anywhere, the
    PositionalEncoding buffer with the reference's values); the training engine -- like every engine -- needs the HIP device."""
    ref = np.load(os.path.join(golden_dir, "c1_abs_pos.npz"))
    from oracle import tcdiff_oracle as O
    m = DanceDecoder(nfeats=151, seq_len=60, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                     cond_feature_dim=438, activation=F.gelu, required_dancer_num=2, use_rotary=False)
    assert sorted(m.state_dict().keys()) == [str(k) for k in ref["state_dict_keys"]]
    sd = O.synth_state_dict(dn=2, seq_len=60, use_rotary=False)
    assert torch.equal(m.state_dict()["abs_pos_encoding.pe"], sd["abs_pos_encoding.pe"])      # (the oracle's = the real module's: generator script)
    m.load_state_dict(sd, strict=True)
    with pytest.raises(L.TcdiffError):            # (no CPU fallback; on the GPU the option trains: test_train_step_gpu.py)
        m.train_engine()


def test_weights_version_sees_in_place_changes_and_replaced_parameters():
    """DanceDecoder._weights_version -- what decides whether an engine's packed weights are current -- walks a CACHED parameter list
    (the module-tree walk is 0.3-0.8 ms in front of every sampler call).  It must still change for an in-place update (optimizer step,
    load_state_dict), for a Parameter object replaced by another one with the same in-place version, and must not change when
    something unrelated is registered elsewhere in the process."""
    import torch.nn as nn
    m = DanceDecoder(nfeats=151, seq_len=60, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1, cond_feature_dim=438,
                     activation=F.gelu, required_dancer_num=2)
    v0 = m._weights_version()
    assert m._weights_version() == v0
    with torch.no_grad():
        next(m.parameters()).add_(1.0)
    v1 = m._weights_version()
    assert v1 != v0
    m.load_state_dict(m.state_dict())                         # copies in place: every version moves
    v2 = m._weights_version()
    assert v2 != v1
    m.final_layer.weight = nn.Parameter(torch.zeros_like(m.final_layer.weight))      # a NEW object
    v3 = m._weights_version()
    assert v3 != v2 and len(v3) == len(v2)
    m.final_layer = nn.Linear(512, 151)                       # a replaced submodule
    v4 = m._weights_version()
    assert v4 != v3
    nn.Linear(3, 3)                                           # unrelated registrations: the list is re-walked, the version is not new
    assert m._weights_version() == v4


def test_steps_per_captured_graph_divide_the_run():
    """GaussianDiffusion._steps_per_graph: a run of equal sampler steps is replayed as whole graphs of u steps; u is the configured
    count when that divides the run, else the nearest divisor (50 DDIM steps: 2 x 25, not 20 + 20 + 10 single steps), and the
    configured count when the run is shorter or has no divisor nearby (what does not fill a graph goes step by step)."""
    f = GaussianDiffusion._steps_per_graph
    assert [f(r, 20) for r in (1000, 900, 100, 90, 50, 45, 20, 10, 1, 37)] == [20, 20, 20, 30, 25, 15, 20, 20, 20, 20]
    for r in range(1, 300):
        u = f(r, 20)
        assert 10 < u <= 30 and (r % u == 0 or u == 20)


def test_trj_dist_raises_as_the_reference_does(small):
    """`trj_dist` (model/model.py:71-97): the reference gathers a [B, H, L, L] bias and adds it to the scores of BOTH attention
    blocks of a layer (model/model.py:326,332) -- the cross-attention's are [B, H, L, S + 2], so the reference raises a RuntimeError
    for any trj_dist at any dancer count (checked against the real module where /root/reference exists).  The drop-in raises a
    RuntimeError as well (before touching a device)."""
    model, diff = small
    trj = torch.randint(0, 10, (1, 450, 450))
    with pytest.raises(RuntimeError, match="trj_dist"):
        model(torch.zeros(1, 450, 151), torch.zeros(1, 301, 438), torch.zeros(1, dtype=torch.long), trj_dist=trj)
    with pytest.raises(RuntimeError, match="trj_dist"):
        diff(torch.zeros(1, 3, 150, 151), torch.zeros(1, 301, 438), trj_dist=trj)
    if not os.path.isdir("/root/reference"):
        return
    from oracle import refload
    from oracle import tcdiff_oracle as O
    ref, _ = refload.build_reference(O.synth_state_dict(dn=2, seq_len=60), dn=2, seq_len=60, n_timestep=100)
    ref.eval()
    with pytest.raises(RuntimeError, match="must match the size"), torch.no_grad():
        ref(torch.stack([O.synth_xT(0, 120)]), torch.stack([O.synth_cond(0, 60)]), torch.tensor([50]), cond_drop_prob=0.0,
            trj_dist=torch.randint(0, 10, (1, 120, 120)))


def test_no_cpu_fallback(small):
    model, diff = small
    with pytest.raises(L.TcdiffError):
        model(torch.zeros(1, 450, 151), torch.zeros(1, 301, 438), torch.zeros(1, dtype=torch.long))
    with pytest.raises(L.TcdiffError):
        diff.p_sample_loop((1, 450, 151), torch.zeros(1, 301, 438))
    with pytest.raises(L.TcdiffError):           # the training loss (forward) is HIP-only too
        diff(torch.zeros(1, 3, 150, 151), torch.zeros(1, 301, 438))


def test_product_code_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tcdiff_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_adan_and_ema_have_no_cpu_fallback():
    """Off-GPU the fused optimizer / EMA raise like every other entry of the package (no second arithmetic path)."""
    import pytest
    import torch
    from tcdiff_amd import Adan
    from tcdiff_amd._lib import TcdiffError
    from tcdiff_amd.diffusion import EMA
    a, b = torch.nn.Linear(7, 5), torch.nn.Linear(7, 5)
    with pytest.raises(TcdiffError):
        EMA(0.9999).update_model_average(a, b)
    for p in a.parameters():
        p.grad = torch.ones_like(p)
    with pytest.raises(TcdiffError):
        Adan(a.parameters(), lr=1e-3).step()
    assert EMA(0.9).update_average(torch.tensor(2.0), torch.tensor(4.0)) == 2.0 * 0.9 + (1 - 0.9) * 4.0   # the formula itself
