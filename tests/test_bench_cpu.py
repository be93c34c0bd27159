"""bench.py's accounting on the CPU: the dispatch window of the profiling child (setup and warm-up excluded from per-step
figures -- ADVICE r5), the roofline arithmetic on a synthetic kernel trace, and the launcher options."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def _trace(n, per_step=("step_prologue_kernel<X>", "gemm_tile_kernel<..>", "chain_kernel<3,4,4>", "sampler_update_kernel(int)")):
    rows, d = [], 0
    for name in ("pack_kernel", "gemm_tile_kernel<setup>", "attention_kernel<music>"):     # checkpoint / job setup
        rows.append((d, name, 1000.0)); d += 1
    for run in range(2):
        rows.append((d, "copyBuffer", 5.0)); d += 1                                       # between the two runs
        for _ in range(n):
            for k in per_step:
                rows.append((d, k, 10.0 if "chain" in k else 1.0)); d += 1
    return rows


def test_window_is_the_second_run_only():
    n = 4
    rows = _trace(n)
    win = bench._window(list(reversed(rows)), n)           # any input order: sorted by dispatch id
    assert len(win) == 4 * n
    assert win[0][1].startswith("step_prologue_kernel") and "sampler_update_kernel" in win[-1][1]
    assert not any("setup" in r[1] or "music" in r[1] or "copyBuffer" in r[1] or "pack" in r[1] for r in win)
    assert sum(v for _, _, v in win) == n * 13.0
    with pytest.raises(RuntimeError):
        bench._window(rows, n + 1)


def test_roofline_from_a_synthetic_trace():
    n = 2
    # one fused launch per layer, 125 us each, nine per step; one attention launch per step => the self-attention counts as chain work
    times = {"void chain_kernel<3, 4, 4>(tcdiff_chain_args)": (9 * n, 9 * n * 125.0),
             "void attention_res_kernel<1>(...)": (n, n * 17.0), "void gemm_tile_kernel<MmaBF16, 0, 2, false>(...)": (2 * n, 2 * n * 20.0)}
    roof, rows = bench.kernel_roofline(times, n, 16, 1, 3, 150, "bf16", timing="synthetic")
    fl = bench.family_flops_per_step(16, 3, 150, sa_in_chain=True)["chain"]
    assert roof["kernel"] == "chain" and roof["timing"] == "synthetic"
    assert abs(roof["achieved"] - fl / (9 * 125.0) / 1e6) < 0.1
    assert abs(roof["frac"] - roof["achieved"] / bench.PEAK_BF16_TFLOPS) < 1e-3
    assert rows["chain"]["launches_per_ddpm_step"] == 9.0 and "traffic_from" not in roof


def test_child_mode_is_an_option_of_the_same_script():
    a = bench.parse(["--child-steps", "40", "--batch", "16"])
    assert a.child_steps == 40 and a.gpus == 1
