#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): sampled clips/sec, 3 dancers x 150 frames, 1000 DDPM steps.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one full `GaussianDiffusion.p_sample_loop` (hoisted music encoder + 1000 guided denoising steps +
the result all-gather) over a batch of 16 synthetic clips per GPU, bf16 MFMA operands.  Inputs (weights, music
features, x_T) are resident in HBM before the timed region.

Ranks.  `--gpus N` with no WORLD_SIZE in the environment makes THIS process a launcher: it never touches the GPU,
starts N fresh child interpreters (tcdiff_amd/launch.py: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, one GPU
each), waits, relays rank 0's JSON line and exits non-zero if any rank failed.  Under `torch.distributed.run`
(WORLD_SIZE already set) the process is a rank; `--gpus` must then equal WORLD_SIZE.  Clips are sharded by global
index, no collective inside the loop, one RCCL all-gather at the end; the timed region is bracketed by barrier +
synchronize and the MAX over ranks is reported.

Rank 0 prints ONE JSON line: whole-job throughput, `roofline` for the kernel with the largest share of GPU time
(algorithmic FLOPs / its average duration INSIDE the running sampler, against the 2.5 PFLOP/s dense bf16 MFMA peak;
the HBM figure is secondary), `parity_mode` (the f32 mode that carries the <= 1e-3 claim, timed here too) and, at
N = 1, `cpu_baseline` (the CPU oracle -- a port of the reference's PyTorch path -- on the host cores).
"""
import argparse
import importlib.util
import json
import re
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_CLIP_STEP = {(3, 150): 55.81, (2, 60): 13.73, (5, 300): 240.60}  # SURVEY.md Appendix B (algorithmic)
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBPS = 8000.0      # HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--batch", type=int, default=16, help="clips per GPU (BASELINE config 2: 16)")
    p.add_argument("--dancers", type=int, default=3)
    p.add_argument("--frames", type=int, default=150)
    p.add_argument("--ddpm-steps", type=int, default=1000)
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "bf16x3"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-parity-mode", action="store_true", help="skip the f32-mode timing")
    p.add_argument("--no-kernel-profile", action="store_true", help="skip the in-sampler per-kernel timing pass")
    p.add_argument("--no-pmc", action="store_true",
                   help="skip the two rocprofv3 PMC child passes (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic live")
    p.add_argument("--cpu-seconds", type=float, default=30.0, help="total CPU-baseline budget (three samples of a third each)")
    p.add_argument("--no-train-step", action="store_true", help="skip the secondary config-5 training-step timing")
    p.add_argument("--no-other-configs", action="store_true", help="skip the config-1 / config-4 sampling lines")
    p.add_argument("--train-batch", type=int, default=32, help="clips per GPU of the training step (BASELINE config 5: 32)")
    p.add_argument("--dump-samples", default=None, help="rank 0 saves the gathered samples of the last job here (tests)")
    p.add_argument("--child-steps", type=int, default=0,
                   help="(internal) profiling child of this script: prepare the sampler, run this many two-branch DDPM steps of the "
                        "top of the schedule twice (warm, measured) and exit -- what the rocprofv3 passes of the parent wrap")
    p.add_argument("--stub", action="store_true",
                   help="rank plumbing only (process group, shard ranges, all-reduce, JSON relay); no GPU work: "
                        "what tests/test_launch_cpu.py runs over gloo")
    return p.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------
# launcher (parent process: no torch, no GPU)
# ----------------------------------------------------------------------------------------------------------------
def _load_launch():
    """tcdiff_amd/launch.py by path: importing the package would import torch, which the parent does not need."""
    spec = importlib.util.spec_from_file_location("_tcdiff_launch", os.path.join(ROOT, "tcdiff_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def launch_ranks(argv, world, timeout=3600.0):
    """Start `world` ranks of this script; relay rank 0's last JSON line.  Returns the exit code."""
    rc, out0, errs = _load_launch().spawn_ranks([os.path.abspath(__file__), *argv], world, timeout=timeout)
    line = None
    for ln in out0.splitlines():
        if ln.startswith("{") and ln.rstrip().endswith("}"):
            line = ln
    if rc != 0 or line is None:
        for r, e in enumerate(errs):
            if e.strip():
                sys.stderr.write(f"---- rank {r} stderr (tail) ----\n{e}\n")
        if rc == 0:
            sys.stderr.write("bench.py: rank 0 printed no JSON line\n")
            rc = 1
        return rc
    print(line, flush=True)
    return 0


# ----------------------------------------------------------------------------------------------------------------
# per-kernel accounting
# ----------------------------------------------------------------------------------------------------------------
def family_flops_per_step(B, dn, S, NL=8, H=8, nf=151, ff=1024, branches=2, sa_in_chain=False):
    """ALGORITHMIC FLOPs of one two-branch DDPM step of B clips, per kernel family (2 FLOP per MAC; the shapes are
    SURVEY.md 2.3 / Appendix B, layer-0 self-attention evaluated once for both branches as the engine does).
    `chain` = the row-block chain launches (csrc/chain.hip): every projection of a layer behind the self-attention
    (fc, w_qs, cross-attention, fc, linear1/2/3, next w_qs/w_ks/w_vs), the final layer (folded into the last linear3) and
    the front launch (last fusion linear + layer-0 QKV); `attention` = self-attention; `gemm_tile` = what is left outside
    the layers (FiLM stack, input projection + fusion linears 1-2); `gemm_rowln` is not launched per step any more.
    sa_in_chain (round 5, the default bf16 path): the self-attention of layers 1.. runs INSIDE the chain launches (chain.hip,
    tcdiff_chain_args.sa_q) -- its FLOPs belong to that family; only layer 0's stays an attention launch."""
    Lq = dn * S
    Rs, R = B * Lq, branches * B * Lq
    M = S + 2
    rowln = 0.0
    tile = 2.0 * Rs * nf * 512 + 2.0 * B * S * 1024 * 512 * dn + 2.0 * B * S * 1024 * 1024   # input proj, f1, f2
    tile += 2.0 * branches * B * 512 * (NL * 3 * 1024)        # FiLM stack
    att, chain = 0.0, 2.0 * R * nf * 512                       # final layer: executed by the last chain launch (folded into
                                                               # its linear3); input projection: by the first fusion GEMM
    chain += 2.0 * B * S * 1024 * 512 * dn + 2.0 * Rs * 1536 * 512   # front launch: last fusion linear + layer-0 QKV
    for l in range(NL):
        nseq_sa = B if l == 0 else branches * B
        sa = 4.0 * nseq_sa * H * Lq * Lq * 64
        if sa_in_chain and l > 0:
            chain += sa
        else:
            att += sa
        chain += 3 * 2.0 * R * 512 * 512 + 2.0 * R * 512 * ff          # fc, fc, linear3, linear2
        chain += 2.0 * R * 512 * 512 + 2.0 * R * ff * 512              # cross-attention w_qs, linear1
        chain += 4.0 * branches * B * H * Lq * M * 64                  # cross-attention
        if l + 1 < NL:
            chain += 2.0 * R * 1536 * 512                              # next layer's w_qs / w_ks / w_vs
    return {"chain": chain, "gemm_rowln": rowln, "gemm_tile": tile, "attention": att}


def family_bytes_per_step(B, dn, S, es, NL=8, H=8, ff=1024, sa_in_chain=False):
    """ALGORITHMIC HBM bytes of one two-branch DDPM step per family at launch granularity: every operand of a launch
    read once, every result written once (SURVEY.md 8(d)'s 110 MB is the figure if no intermediate ever left the chip).
    Chain launch: O in, x in / out (fp32), Q / K / V images out, the layer's weights and K / V caches once.
    sa_in_chain: layers 1.. read the previous launch's Q / K / V (3 x R x 512) instead of O (R x 512); one attention launch
    (layer 0: one branch's Q / K / V in, O out) is left."""
    Lq = dn * S
    R = 2 * B * Lq
    act = lambda cols, e=es: R * cols * e
    x32 = R * 512 * 4
    lpc = (S + 2 + 31) // 32 * 32
    wl = (5 * 512 * 512 + 2 * 512 * ff + 1536 * 512) * es                 # weights of one chain launch
    chain = NL * (act(512) + 2 * x32 + 3 * act(512) + wl + 2 * (B + 1) * H * lpc * 64 * es)
    att = NL * (3 * act(512) + act(512))
    if sa_in_chain:
        chain += (NL - 1) * 2 * act(512)
        att = (3 * act(512) + act(512)) // 2
    tile = (act(512) + 1536 * 512 * es + 3 * act(512)) // 2
    return {"chain": chain, "gemm_rowln": 0.0, "gemm_tile": tile, "attention": att}


FAMILIES = {"chain": ("chain_kernel",), "gemm_rowln": ("gemm_rowln_kernel",), "gemm_tile": ("gemm_tile_kernel",),
            "attention": ("attention_res_kernel", "attention_kernel")}


def insampler_kernel_times(diff, shape, cond, xT, n_steps=40):
    """Per-kernel device time INSIDE the running sampler (both free-running streams, graph replay), measured live with
    torch.profiler (roctracer activity records) over `n_steps` two-branch DDPM steps.  Returns
    {kernel name: (launches, total_us)}, steps profiled."""
    import torch
    from torch.profiler import ProfilerActivity, profile
    from tcdiff_amd import _lib as L
    T = diff.n_timestep
    tseq = list(range(T - 1, T - 1 - n_steps, -1))
    run = lambda: diff._run(L.SAMPLER_DDPM, tuple(shape), cond, xT.float(), tseq, diff._ddpm_params(tseq), seed=7)
    run()                                  # graphs of this key are captured by the timed jobs already; settle
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        run()
        torch.cuda.synchronize()
    out = {}
    for ev in prof.key_averages():
        dt = getattr(ev, "self_device_time_total", None)
        if dt is None:
            dt = getattr(ev, "self_cuda_time_total", 0.0)
        if dt and ev.count:
            out[ev.key] = (int(ev.count), float(dt))
    return out, n_steps


def kernel_roofline(times, n_steps, B_launch, streams, dn, S, dtype, timing=None):
    """Fold profiler records into families; the dominant family's achieved rate = its algorithmic FLOPs per DDPM step /
    its device time per DDPM step."""
    peak = PEAK_BF16_TFLOPS if dtype == "bf16" else PEAK_F32_TFLOPS
    es = 2 if dtype == "bf16" else 4
    B = B_launch * streams
    # the self-attention of layers 1.. runs inside the chain launches when the profile shows one attention launch per step
    n_att = sum(c for n, (c, _) in times.items() if any(p in n for p in FAMILIES["attention"])) / max(n_steps, 1)
    sa_in_chain = n_att < 4
    flops = family_flops_per_step(B, dn, S, sa_in_chain=sa_in_chain)
    nbytes = family_bytes_per_step(B, dn, S, es, sa_in_chain=sa_in_chain)
    fam = {}
    total_us = sum(t for _, t in times.values())
    for name, (cnt, us) in times.items():
        for f, pats in FAMILIES.items():
            if any(p in name for p in pats):
                c0, u0 = fam.get(f, (0, 0.0))
                fam[f] = (c0 + cnt, u0 + us)
    rows = {}
    for f, (cnt, us) in fam.items():
        per_step_us = us / n_steps
        rows[f] = dict(launches_per_ddpm_step=round(cnt / n_steps, 1), avg_launch_ms=round(us / cnt / 1e3, 5),
                       device_ms_per_ddpm_step=round(per_step_us / 1e3, 4),
                       share_of_gpu_time=round(us / total_us, 4),
                       tflops=round(flops[f] / per_step_us / 1e6, 1), mfma_frac=round(flops[f] / per_step_us / 1e6 / peak, 4),
                       algo_gbps=round(nbytes[f] / per_step_us / 1e3, 1),
                       hbm_frac=round(nbytes[f] / per_step_us / 1e3 / PEAK_HBM_GBPS, 4))
    if not rows:
        return None, rows
    dom = max(rows, key=lambda k: rows[k]["device_ms_per_ddpm_step"])
    d = rows[dom]
    roof = dict(bound="mfma", kernel=dom, achieved=d["tflops"], peak=peak, unit="TFLOP/s", frac=d["mfma_frac"],
                traffic=None, avg_launch_ms=d["avg_launch_ms"], launches_per_ddpm_step=d["launches_per_ddpm_step"],
                share_of_gpu_time=d["share_of_gpu_time"],
                timing=timing or (f"torch.profiler device durations inside the running sampler, {n_steps} two-branch DDPM "
                                  f"steps, {streams} free-running stream(s) of {B_launch} clips"),
                algorithmic_gflop_per_ddpm_step=round(flops[dom] / 1e9, 2),
                hbm=dict(algorithmic_gbps=d["algo_gbps"], peak=PEAK_HBM_GBPS, frac=d["hbm_frac"],
                         algorithmic_mb_per_ddpm_step=round(nbytes[dom] / 1e6, 1)))
    if dom == "chain":
        # The launch streams every row block's weights L2 -> CU (a CU cannot hold more than ~64 of the 14 400 rows).  Round 4
        # measured what that costs and what it does not (profiles/r04_chain_experiments.txt): with pure loads EVERY CU streams
        # the same 5.5 MB at 113-117 GB/s (26-29 TB/s chip-wide) at 1, 225 and 256 blocks -- the per-CU vector-memory return
        # path, not the L2, is the ceiling of the stream -- and the chain launch's slow-down from 1 to 225 blocks is the shader
        # clock (2.39 -> 1.9-2.0 GHz at the power limit), not contention.  Reported beside the MFMA figure.
        # (row blocks are cut per sequence when the self-attention runs inside the launch: 8 blocks per 450-row sequence)
        n_blk = 2 * B * ((dn * S + 63) // 64) if sa_in_chain else (2 * B * dn * S + 63) // 64
        per_step = (7 * 176 + 128) * 4096 * 8 * n_blk + 80 * 4096 * 8 * ((B * S + 63) // 64) * dn
        gbps = per_step / (d["device_ms_per_ddpm_step"] * 1e-3) / 1e9
        roof["weight_stream"] = dict(
            bytes_per_ddpm_step=int(per_step), achieved_gbps=round(gbps, 1), achieved_gbps_per_cu=round(gbps / n_blk, 1),
            probe_ceiling_gbps_per_cu=115.0, frac_of_probe_ceiling=round(gbps / n_blk / 115.0, 3),
            note="L2 -> CU weight bytes of the chain launches (row blocks x stream bytes) / their device time, epilogues included; "
                 "ceiling = tools/probe/l2_alias_probe.hip, pure loads of the same stream by 225 / 256 CUs at once "
                 "(profiles/r04_chain_experiments.txt)")
    return roof, rows


# ----------------------------------------------------------------------------------------------------------------
# launch durations and HBM-side traffic, measured by THIS run: three rocprofv3 passes over a short child run of this script
# ----------------------------------------------------------------------------------------------------------------
PMC_FAMILIES = {"chain": "chain_kernel", "gemm_rowln": "gemm_rowln_kernel", "gemm_tile": "gemm_tile_kernel", "attention": "attention"}


def _run_profiler_child(cmd, env, timeout):
    """One rocprofv3 pass in its OWN process group, output to temp files (no pipes a grandchild could keep open).  On a timeout
    the whole group is killed and reaped before this returns: an orphaned sampler must not run beside the timings that follow."""
    import signal
    import subprocess
    import tempfile
    with tempfile.TemporaryFile(mode="w+", dir="/tmp") as fo, tempfile.TemporaryFile(mode="w+", dir="/tmp") as fe:
        p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=fo, stderr=fe, start_new_session=True)
        try:
            rc = p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
            p.wait()
            raise RuntimeError(f"profiler pass timed out after {timeout:.0f} s (process group killed)")
        fe.seek(0)
        return rc, fe.read()[-300:]


def _window(rows, n):
    """The dispatches of the child's SECOND run of n DDPM steps: from the first step_prologue behind the n-th sampler_update
    through the 2n-th sampler_update (rows = [(dispatch id, kernel name, value)] in dispatch order).  Setup kernels (weight
    packing, the music encoder, the FiLM table) and the warm run lie outside."""
    rows = sorted(rows, key=lambda r: r[0])
    upd = [i for i, r in enumerate(rows) if "sampler_update_kernel" in r[1]]
    if len(upd) < 2 * n:
        raise RuntimeError(f"{len(upd)} sampler_update launches in the trace, expected {2 * n}")
    lo = next(i for i in range(upd[n - 1] + 1, len(rows)) if "step_prologue_kernel" in rows[i][1])
    return rows[lo:upd[2 * n - 1] + 1]


def live_child_passes(a, n=40, timeout=300.0):
    """Three rocprofv3 passes -- `--kernel-trace` alone (launch durations), `--kernel-trace --pmc FETCH_SIZE`, `--kernel-trace --pmc
    WRITE_SIZE` (separate passes, as MI355X_MICROARCH.md prescribes: the two counters do not fit one pass, and counter collection
    perturbs durations) -- around `python3 bench.py --child-steps n`, started as CHILDREN of this process (never an exec: this
    process has the GPU open).  The child runs n two-branch DDPM steps of the top of the schedule twice on the same build, shapes
    and batch as the timed run; only the second run's dispatches are counted.  Returns
    {"times": {kernel name: (launches, total_us)}, "n": n, "bytes_per_launch": {family: bytes}, "bytes_per_step": bytes}; bytes
    carry the guide's gfx950 correction (FETCH_SIZE counts 128-byte requests as 64: doubled; both are in KB)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        raise RuntimeError("rocprofv3 not found")
    child = [sys.executable if os.path.basename(sys.executable).startswith("python") else "python3", os.path.abspath(__file__),
             "--child-steps", str(n), "--ddpm-steps", str(a.ddpm_steps), "--batch", str(a.batch), "--dancers", str(a.dancers),
             "--frames", str(a.frames), "--dtype", a.dtype]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = {"n": n, "errors": {}}
    for what in ("trace", "FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="tcdiff_prof_", dir="/tmp")
        try:
            extra = [] if what == "trace" else ["--pmc", what]
            rc, err = _run_profiler_child([exe, "--kernel-trace"] + extra + ["--output-format", "csv", "-d", d, "--"] + child, env, timeout)
            pat = "*kernel_trace.csv" if what == "trace" else "*counter_collection.csv"
            files = glob.glob(os.path.join(d, "**", pat), recursive=True)
            if rc != 0 or not files:
                raise RuntimeError(f"rocprofv3 {what}: rc {rc}, {len(files)} files: {err}")
            rows = []
            for f in files:
                for row in csv.DictReader(open(f)):
                    if what == "trace":
                        rows.append((int(row["Dispatch_Id"]), row["Kernel_Name"],
                                     (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3))
                    elif row.get("Counter_Name") == what:
                        rows.append((int(row["Dispatch_Id"]), row.get("Kernel_Name", ""), float(row["Counter_Value"])))
            win = _window(rows, n)
            if what == "trace":
                times = {}
                for _, name, us in win:
                    c0, u0 = times.get(name, (0, 0.0))
                    times[name] = (c0 + 1, u0 + us)
                out["times"] = times
            else:
                fam = {k: [0.0, 0] for k in PMC_FAMILIES}
                for _, name, v in win:
                    for k, sym in PMC_FAMILIES.items():
                        if sym in name.split("(")[0]:
                            fam[k][0] += v
                            fam[k][1] += 1
                out[what] = (sum(v for _, _, v in win), fam)
        except Exception as e:
            out["errors"][what] = repr(e)[:300]
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
        (ft, ff), (wt, wf) = out.pop("FETCH_SIZE"), out.pop("WRITE_SIZE")
        out["bytes_per_step"] = int((2 * ft + wt) * 1024 / n)
        out["bytes_per_launch"] = {k: int((2 * ff[k][0] / ff[k][1] + wf[k][0] / wf[k][1]) * 1024)
                                   for k in PMC_FAMILIES if ff[k][1] and wf[k][1]}
    return out


def profiling_child(a, diff, shape, cond, xT):
    """`--child-steps n`: what the parent's rocprofv3 passes wrap."""
    import torch
    from tcdiff_amd import _lib as L
    T = diff.n_timestep
    tseq = list(range(T - 1, T - 1 - a.child_steps, -1))
    for _ in range(2):
        diff._run(L.SAMPLER_DDPM, tuple(shape), cond, xT.float(), tseq, diff._ddpm_params(tseq), seed=7)
        torch.cuda.synchronize()


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline
# ----------------------------------------------------------------------------------------------------------------
def host_cpu():
    model, cores = "unknown", set()
    try:
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                phys = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":", 1)[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return model, len(cores) or None, os.cpu_count()


def cpu_baseline(dn, S, T, seconds):
    """The CPU oracle (port of the reference's PyTorch path) on the host cores: guided DDPM steps of ONE clip,
    run for ~`seconds`, extrapolated to T steps."""
    import torch
    from oracle import tcdiff_oracle as O
    sd = O.synth_state_dict(dn=dn, seq_len=S)
    cond = torch.stack([O.synth_cond(0, S)])
    x = torch.stack([O.synth_xT(0, dn * S)])
    tab = O.make_tables(T)
    model, phys, logical = host_cpu()
    # pick the intra-op thread count that is fastest on this host (all logical CPUs oversubscribes badly)
    ncpu = logical or 1
    best = (None, 1e30)
    with torch.no_grad():
        for th in sorted({min(ncpu, c) for c in (8, 16, 32, 64)}):
            torch.set_num_threads(th)
            t1 = time.time()
            O.p_sample(sd, tab, x, cond, T - 1, T, 2, torch.randn(x.shape))
            el = time.time() - t1
            if el < best[1]:
                best = (th, el)
            if el > 20:
                break
    torch.set_num_threads(best[0])
    # three samples of seconds / 3 each; the MEDIAN is reported and the spread printed beside it (a shared host: one
    # sample of 12 s differed by 24 % between two runs of round 2)
    rates, n_tot, t_tot, i = [], 0, 0.0, T - 1
    with torch.no_grad():
        for _ in range(3):
            n, t0 = 0, time.time()
            while True:
                x, _ = O.p_sample(sd, tab, x, cond, i, T, 2, torch.randn(x.shape))
                n, i = n + 1, (i - 1) % T
                if time.time() - t0 >= seconds / 3:
                    break
            dt = time.time() - t0
            rates.append(1.0 / (dt / n * T))
            n_tot, t_tot = n_tot + n, t_tot + dt
    rates.sort()
    return dict(value=round(rates[1], 6), unit="clips/s", cores=torch.get_num_threads(), kind="port",
                cpu_model=model, physical_cores=phys, logical_cpus=logical,
                samples=[round(r, 6) for r in rates], spread=round((rates[2] - rates[0]) / rates[1], 3),
                sample=f"median of 3 samples, {n_tot} guided DDPM steps of 1 clip ({dn} dancers x {S} frames) in all, on the "
                       f"CPU oracle (torch CPU fp32, {torch.get_num_threads()} threads), {t_tot:.1f} s, each extrapolated to {T} steps")


# ----------------------------------------------------------------------------------------------------------------
# rank body
# ----------------------------------------------------------------------------------------------------------------
def rank_facts(D, n_total, rank, world, dev):
    """(ranks_seen, [[lo, hi] per rank]): one all-reduce and one all-gather over the job's process group (RCCL on GPUs)."""
    import torch
    import torch.distributed as dist
    lo, hi = D.shard_range(n_total, rank, world)
    if not D.collectives_on():
        return 1, [[lo, hi]]
    if dist.get_backend() == "gloo":
        dev = "cpu"
    one = torch.ones(1, device=dev, dtype=torch.int32)
    dist.all_reduce(one)
    mine = torch.tensor([lo, hi], device=dev, dtype=torch.int32)
    allr = torch.empty(world * 2, device=dev, dtype=torch.int32)
    dist.all_gather_into_tensor(allr, mine)
    return int(one.item()), allr.view(world, 2).tolist()


def stub_rank(a):
    """Rank plumbing without GPU work (CPU test of the launcher, gloo)."""
    import torch
    from tcdiff_amd import dist as D
    rank, world, local = D.init_from_env("gloo" if not torch.cuda.is_available() else None)
    n_total = a.batch * world
    seen, ranges = rank_facts(D, n_total, rank, world, "cpu")
    t = D.max_over_ranks(0.001 * (rank + 1), "cpu")
    D.barrier()
    if rank == 0:
        print(json.dumps({"metric": "stub", "n_gpus": world, "ranks_seen": seen, "clip_ranges": ranges,
                          "local_rank": local, "max_time": t}), flush=True)
    else:
        print(json.dumps({"rank": rank, "local_rank": local}), flush=True)   # not relayed: only rank 0's stdout is
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def rank_main(a):
    import torch
    import torch.nn.functional as F
    from tcdiff_amd import dist as D
    # TCDIFF_BENCH_ONE_DEVICE=1 (tests): every rank runs on GPU 0 and the collectives go through gloo -- the rank plumbing of a
    # multi-GPU run (sharding, clip offsets, gather, max-over-ranks timing) on a one-GPU box; RCCL refuses duplicate devices
    one_dev = os.environ.get("TCDIFF_BENCH_ONE_DEVICE", "0") == "1"
    rank, world, local = D.init_from_env("gloo" if one_dev else None)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    if one_dev:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from tcdiff_amd import DanceDecoder, GaussianDiffusion
    from tcdiff_amd import weights as W

    dn, S, T, B = a.dancers, a.frames, a.ddpm_steps, a.batch
    Lq = dn * S

    def build(compute):
        model = DanceDecoder(nfeats=151, seq_len=S, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8,
                             dropout=0.1, cond_feature_dim=438, activation=F.gelu, required_dancer_num=dn,
                             compute_dtype=compute)
        model.load_state_dict(W.synth_state_dict_like(model))
        model.eval()
        diff = GaussianDiffusion(model, S, 151, None, schedule="cosine", n_timestep=T, predict_epsilon=False,
                                 loss_type="l2", use_p2=False, cond_drop_prob=0.25, guidance_weight=2, seq_len=S)
        return diff.to(dev).eval()

    diff = build(a.dtype)
    n_total = B * world
    lo, hi = D.shard_range(n_total, rank, world)
    cond = torch.stack([W.synth_cond(c, S) for c in range(lo, hi)]).to(dev)
    xT = torch.stack([W.synth_xT(c, Lq) for c in range(lo, hi)]).to(dev)
    ranks_seen, clip_ranges = rank_facts(D, n_total, rank, world, dev)
    xT0, cond0 = W.synth_xT(0, Lq)[None].to(dev), W.synth_cond(0, S)[None].to(dev)       # clip 0 (the goldens' clip)

    def one_job():
        x = diff.p_sample_loop((hi - lo, Lq, 151), cond, noise=xT, seed=1234, clip_offset=lo)
        return D.gather_samples(x, n_total)

    if a.child_steps > 0:
        profiling_child(a, diff, (hi - lo, Lq, 151), cond, xT)
        return
    for _ in range(a.warmup):
        one_job()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = one_job()
    torch.cuda.synchronize()
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0, dev)
    assert out.shape == (n_total, Lq, 151) and bool(torch.isfinite(out).all())
    if a.dump_samples and rank == 0:
        torch.save(out.cpu(), a.dump_samples)

    if rank == 0:
        clips_per_s = n_total * a.steps / dt
        gf = GFLOP_PER_CLIP_STEP.get((dn, S))
        nb = hi - lo
        streams = 1
        res = {
            "metric": "sampled clips/sec (3 dancers x 150 frames, 1000 DDPM steps)",
            "value": round(clips_per_s, 4), "unit": "clips/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"batch={B} clips/GPU, {dn} dancers x {S} frames, {T} DDPM steps (p_sample_loop, "
                                   f"CFG w=2, cosine schedule), {a.dtype}, {world}xMI355X",
                       "clips_per_gpu": B, "ddpm_steps": T, "tokens_per_clip": Lq},
            "ranks_seen": ranks_seen, "clip_ranges": clip_ranges,
            "ms_per_ddpm_step": round(dt / a.steps / T * 1e3, 4),
        }
        if gf is not None:
            peak = PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS
            # algorithmic: SURVEY.md Appendix B, both CFG branches on every step.  executed: the last 10 % of the steps have
            # guidance weight 1 (model/diffusion.py:219-224) and run the conditional branch only, and layer 0's
            # self-attention is evaluated once for both branches.  `mfma_frac_per_gpu` uses the ALGORITHMIC figure (the one
            # BASELINE's 50 clips/s target is priced with); `executed_mfma_frac_per_gpu` what the MFMA pipes actually did.
            fl = family_flops_per_step(1, dn, S)
            two, one = sum(fl.values()) / 1e9, sum(family_flops_per_step(1, dn, S, branches=1).values()) / 1e9
            n_one = sum(1 for i in range(T) if i < 0.1 * T)
            gf_exec = (two * (T - n_one) + one * n_one) / T
            res["whole_path"] = {"algorithmic_gflop_per_clip_step": gf,
                                 "executed_gflop_per_clip_step": round(gf_exec, 2),
                                 "achieved_tflops_per_gpu": round(clips_per_s / world * gf * T / 1e3, 2),
                                 "mfma_frac_per_gpu": round(clips_per_s / world * gf * T / 1e3 / peak, 4),
                                 "mfma_frac_uses": "algorithmic_gflop_per_clip_step",
                                 "executed_tflops_per_gpu": round(clips_per_s / world * gf_exec * T / 1e3, 2),
                                 "executed_mfma_frac_per_gpu": round(clips_per_s / world * gf_exec * T / 1e3 / peak, 4)}
        roof, rows = None, {}
        if not a.no_kernel_profile and T >= 200:
            # Launch durations and HBM-side bytes from rocprofv3 passes over a child run of this very command (the same files a
            # `rocprofv3 --kernel-trace --stats` of it gives: profiles/rNN_kernel_stats_*): `frac` is reproducible from them and the
            # family sum fits the wall time.  torch.profiler inside this process is the fallback (it inflates launches by 4-5 %).
            prof = None
            if world == 1 and not a.no_pmc and a.dtype == "bf16":
                try:
                    prof = live_child_passes(a)
                except Exception as e:
                    res["profiler_child_error"] = repr(e)[:300]
            try:
                if prof and "times" in prof:
                    roof, rows = kernel_roofline(
                        prof["times"], prof["n"], nb // streams, streams, dn, S, a.dtype,
                        timing=f"rocprofv3 --kernel-trace pass over a child run of this command: the second of two runs of "
                               f"{prof['n']} two-branch DDPM steps, {nb} clips")
                else:
                    times, n_prof = insampler_kernel_times(diff, (nb, Lq, 151), cond, xT)
                    roof, rows = kernel_roofline(times, n_prof, nb // streams, streams, dn, S, a.dtype)
            except Exception as e:   # the throughput line must not depend on the profiler
                res["kernel_profile_error"] = repr(e)[:300]
            if roof and prof:
                dom = roof["kernel"] if roof["kernel"] in PMC_FAMILIES else "chain"
                if "bytes_per_launch" in prof and dom in prof["bytes_per_launch"]:
                    roof["traffic"] = prof["bytes_per_launch"][dom]
                    roof["traffic_from"] = (f"this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE child passes, the same "
                                            f"{prof['n']} DDPM steps (FETCH_SIZE doubled, KB -> bytes)")
                    roof["traffic_per_ddpm_step"] = {"bytes": prof["bytes_per_step"], "algorithmic_min_bytes": 110e6,
                                                     "from": f"this run: every dispatch of those {prof['n']} steps (prologue "
                                                             f"through sampler update; setup and warm-up excluded)"}
                if prof.get("errors"):
                    roof["profiler_pass_errors"] = prof["errors"]
        res["roofline"] = roof
        res["kernels"] = rows
        if world == 1 and not a.no_parity_mode and a.dtype == "bf16" and T >= 200:
            # the two modes held to <= 1e-3 against the reference, timed on the same workload: "f32" (v_mfma_f32_32x32x2_f32, an
            # exact fp32 fma chain) and "bf16x3" (fp32 storage, every product as three bf16 MFMAs on (hi, lo) splits)
            pm = {}
            for mode in ("f32", "bf16x3"):
                dm = build(mode)
                dm.p_sample_loop((nb, Lq, 151), cond, noise=xT, seed=1234, start_point=int(0.1 * T) + 4)   # warm + capture
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                dm.p_sample_loop((nb, Lq, 151), cond, noise=xT, seed=1234)
                torch.cuda.synchronize()
                d1 = time.perf_counter() - t1
                pm[mode] = {"dtype": mode, "value": round(nb / d1, 4), "unit": "clips/s", "ms_per_step": round(d1 * 1e3, 1),
                            "sample": f"one full job of {nb} clips x {T} steps",
                            "tolerance": "max-abs <= 1e-3 vs the reference on fp32 (tests/test_parity_gpu.py)"}
                del dm
            pm["f32"]["mfma_frac_f32_peak"] = round(pm["f32"]["value"] * gf * T / 1e3 / PEAK_F32_TFLOPS, 4) if gf else None
            # three bf16 MFMAs per product: executed matrix work = 3 x the algorithmic FLOPs, priced against the bf16 peak
            pm["bf16x3"]["mfma_frac_bf16_peak_executed"] = \
                round(3 * pm["bf16x3"]["value"] * gf * T / 1e3 / PEAK_BF16_TFLOPS, 4) if gf else None
            res["parity_mode"] = dict(pm["f32"], fast=pm["bf16x3"])
        if world == 1 and (dn, S, T) == (3, 150, 1000):
            # how far the BENCHMARKED arithmetic is from the reference: one guided evaluation of clip 0 at two timesteps against
            # the committed outputs of the real reference on the same name-keyed synthetic weights and inputs
            # (tests/golden/c2_forward.npz, made by tests/golden/make_golden.py); asserted in tests/test_parity_gpu.py
            try:
                import numpy as np
                gz = np.load(os.path.join(ROOT, "tests", "golden", "c2_forward.npz"))
                errs = {}
                for t in (999, 37):
                    y = diff.model.guided_forward(xT0[:1], cond0[:1], torch.full((1,), t, dtype=torch.long, device=dev), 2)
                    errs[f"t{t}"] = float(np.abs(y.detach().cpu().double().numpy() - gz[f"guided_w2_t{t}"]).max())
                res[f"{a.dtype}_vs_reference_maxabs"] = dict(
                    guided_evaluation=round(max(errs.values()), 6), per_timestep={k: round(v, 6) for k, v in errs.items()},
                    reference="real reference (fp32 CPU), clip 0, w = 2; outputs are O(1)",
                    bound_in_tests=2.5e-2 if a.dtype == "bf16" else 5e-4)
            except Exception as e:
                res["vs_reference_error"] = repr(e)[:300]
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(dn, S, T, a.cpu_seconds)
        if world == 1 and not a.no_other_configs and (dn, S, T) == (3, 150, 1000):
            # BASELINE.json's other single-GPU sampling configurations, one timed job each after a warm-up job (they are
            # parity-test cases, not the metric: reported so that a driver run sees what DESIGN.md section 5 quotes)
            try:
                res["other_configs"] = other_configs(a, dev)
            except Exception as e:
                res["other_configs_error"] = repr(e)[:300]
    # ---- secondary: the training step (BASELINE config 5: batch 32 per GPU, Adan, data-parallel over the job's ranks) ----------
    ts, ts_err = None, None
    if not a.no_train_step:
        try:
            ts = train_step_bench(a, D, dev, world, dn, S)
        except Exception as e:           # the headline line must not depend on the secondary measurement
            ts_err = repr(e)[:300]
    if rank == 0:
        if ts is not None:
            res["train_step"] = ts
        if ts_err is not None:
            res["train_step_error"] = ts_err
        print(json.dumps(res), flush=True)
    D.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def other_configs(a, dev):
    import torch
    import torch.nn.functional as F
    from tcdiff_amd import DanceDecoder, GaussianDiffusion
    from tcdiff_amd import weights as W
    out = {}
    # (name, dancers, frames, T, clips, sampler): config 1 and 4 of BASELINE.json with p_sample_loop, and what TCDiff.py
    # itself calls when it renders -- ddim_sample, 50 steps, a handful of 3 x 150 clips with the trajectory in-painted
    # (TCDiff.py:292-303, model/diffusion.py:386-442): the small-batch latency regime
    for name, dn, S, T, nb, sampler in (("config1_1clip_2x60_T100", 2, 60, 100, 1, "ddpm"),
                                        ("config4_5x300_T1000_batch4", 5, 300, 1000, 4, "ddpm"),
                                        ("ddim50_3x150_1clip", 3, 150, 1000, 1, "ddim"),
                                        ("ddim50_3x150_4clips", 3, 150, 1000, 4, "ddim")):
        model = DanceDecoder(nfeats=151, seq_len=S, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                             cond_feature_dim=438, activation=F.gelu, required_dancer_num=dn, compute_dtype=a.dtype)
        model.load_state_dict(W.synth_state_dict_like(model))
        diff = GaussianDiffusion(model.eval(), S, 151, None, schedule="cosine", n_timestep=T, predict_epsilon=False,
                                 loss_type="l2", use_p2=False, cond_drop_prob=0.25, guidance_weight=2, seq_len=S).to(dev).eval()
        Lq = dn * S
        cond = torch.stack([W.synth_cond(c, S) for c in range(nb)]).to(dev)
        xT = torch.stack([W.synth_xT(c, Lq) for c in range(nb)]).to(dev)
        if sampler == "ddim":
            x0 = torch.stack([W.synth_xT(100 + c, Lq, 3) for c in range(nb)]).clamp(-1, 1).to(dev)
            job = lambda: diff.ddim_sample((nb, Lq, 151), cond, x_0=x0, init_noise=xT, seed=1)
            steps = 50
        else:
            job = lambda: diff.p_sample_loop((nb, Lq, 151), cond, noise=xT, seed=1)
            steps = T
        job()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = job()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        assert bool(torch.isfinite(x).all())
        gf = GFLOP_PER_CLIP_STEP.get((dn, S))
        out[name] = {"value": round(nb / dt, 4), "unit": "clips/s", "clips": nb, "dancers": dn, "frames": S, "sampler": sampler,
                     "steps": steps, "dtype": a.dtype, "ms_per_job": round(dt * 1e3, 1), "ms_per_step": round(dt * 1e3 / steps, 4),
                     "mfma_frac": round(nb / dt * gf * steps / 1e3 / PEAK_BF16_TFLOPS, 4) if (gf and a.dtype == "bf16") else None}
        del diff, model
    return out


def train_step_bench(a, D, dev, world, dn, S, iters=10, warm=5):
    """TCDiff.train_loop's body (TCDiff.py:227-245) on every rank: diffusion(x, cond) in train mode (dropout live), zero_grad,
    backward (gradients averaged over the ranks by RCCL all-reduces launched layer by layer under the backward), fused Adan,
    fused EMA.  `value` = whole-job steps per second of the data-parallel job = 1 / (max-over-ranks time per step)."""
    import torch
    import torch.nn.functional as F
    from tcdiff_amd import Adan, DanceDecoder, GaussianDiffusion
    from tcdiff_amd import weights as W
    b = a.train_batch
    model = DanceDecoder(nfeats=151, seq_len=S, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                         cond_feature_dim=438, activation=F.gelu, required_dancer_num=dn, compute_dtype=a.dtype)
    model.load_state_dict(W.synth_state_dict_like(model))
    diff = GaussianDiffusion(model, S, 151, None, schedule="cosine", n_timestep=a.ddpm_steps, predict_epsilon=False,
                             loss_type="l2", use_p2=False, cond_drop_prob=0.25, guidance_weight=2, seq_len=S).to(dev)
    diff.train()
    optim = Adan(model.parameters(), lr=5e-5, weight_decay=0.02)
    if D.collectives_on():
        model.train_engine().enable_grad_sync()            # data-parallel gradient averaging is opt-in (dist.FlatGradientAllReducer)
    g = torch.Generator().manual_seed(4242 + int(os.environ.get("RANK", "0")))
    x = (torch.rand(b, dn, S, 151, generator=g) * 2 - 1).to(dev)
    cond = torch.randn(b, 2 * S + 1, 438, generator=g).to(dev)

    def step():
        total, _ = diff(x, cond)
        optim.zero_grad()
        total.backward()
        optim.step()
        diff.ema.update_model_average(diff.master_model, diff.model)
        return total
    for _ in range(warm):
        step()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        last = step()
    torch.cuda.synchronize()
    D.barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0, dev) / iters
    fl = 3 * (GFLOP_PER_CLIP_STEP.get((dn, S), 0.0) / 2 + 1.53) * 1e9 * b      # forward (one conditional evaluation + music branch) x 3
    eng = model.train_engine()
    sync = eng.grad_sync
    return dict(metric="training steps/sec (config 5: forward with dropout + 4-term loss + backward + Adan + EMA)",
                value=round(1.0 / dt, 3), unit="steps/s", ms_per_step=round(dt * 1e3, 3), batch_per_gpu=b, global_batch=b * world,
                clips_per_s=round(b * world / dt, 1), dtype=a.dtype, n_gpus=world, steps=iters, warmup=warm,
                loss=round(float(last), 5), algorithmic_tflop_per_gpu_step=round(fl / 1e12, 3),
                tflops_per_gpu=round(fl / dt / 1e12, 1), mfma_frac_per_gpu=round(fl / dt / 1e12 / PEAK_BF16_TFLOPS, 4),
                grad_allreduce=(dict(collectives_per_step=sync.launched // (iters + warm), mb_per_step=round(eng.n_grad * 4 / 1e6, 1),
                                     overlap="one async RCCL all-reduce per decoder layer, launched as its gradients complete")
                                if (sync and D.collectives_on()) else None))


def main():
    a = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and a.gpus > 1:
        sys.exit(launch_ranks(sys.argv[1:], a.gpus))          # parent: spawns, relays, never touches the GPU
    world = int(env_world) if env_world is not None else 1
    if world != a.gpus:
        sys.stderr.write(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}: launch with "
                         f"`python bench.py --gpus {a.gpus}` (it starts the ranks itself) or make them agree\n")
        sys.exit(2)
    if a.stub:
        stub_rank(a)
    else:
        rank_main(a)


if __name__ == "__main__":
    main()
