#!/usr/bin/env python3
"""Headline benchmark (BASELINE.json): sampled clips/sec, 3 dancers x 150 frames, 1000 DDPM steps.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one full `GaussianDiffusion.p_sample_loop` (hoisted music encoder + 1000 guided denoising steps +
the result all-gather) over a batch of 16 synthetic clips per GPU, bf16 MFMA operands.  Inputs (weights, music
features, x_T) are resident in HBM before the timed region.  For N > 1 the driver launches one process per GPU
(torch.distributed.run); clips are sharded by global index, no collective inside the loop, one RCCL all-gather
at the end; the timed region is bracketed by barrier + synchronize and the MAX over ranks is reported.

Rank 0 prints ONE JSON line with the whole-job throughput, a `roofline` object for the dominant kernel
(algorithmic FLOPs / measured HIP-event duration, against the 2.5 PFLOP/s dense bf16 MFMA peak) and, at N=1,
a `cpu_baseline` object (the CPU oracle -- a port of the reference's PyTorch path -- timed on the host cores on a
bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

GFLOP_PER_CLIP_STEP = {(3, 150): 55.81, (2, 60): 13.73, (5, 300): 240.60}  # SURVEY.md Appendix B (algorithmic)
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBPS = 8000.0      # HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured achievable


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=2)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--batch", type=int, default=16, help="clips per GPU (BASELINE config 2: 16)")
    p.add_argument("--dancers", type=int, default=3)
    p.add_argument("--frames", type=int, default=150)
    p.add_argument("--ddpm-steps", type=int, default=1000)
    p.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    return p.parse_args()


def event_time_ms(fn, iters=20, warm=3):
    """Average duration of fn() measured with HIP events on the stream the kernels are launched on."""
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


def kernel_roofline(eng, B, dtype, streams=1):
    """Per-kernel achieved rate at the shapes the sampler LAUNCHES: with `streams` free-running sub-batches of B clips
    each (tcdiff_amd/diffusion.py, dual_parts) one launch covers R = 2*B*L rows and every kernel is launched `streams`
    times per DDPM step, two launches of different kernels running side by side.  Returns (dominant, all)."""
    from tcdiff_amd import _lib as L
    from tcdiff_amd import kernels as K
    dt, w, b = eng.dt, eng.w, eng.b
    Lq, S, H = eng.Lseq, eng.S, eng.H
    R = 2 * B * Lq
    NLy = eng.NL
    fld = NLy * 3 * 1024
    rope = w["rope"]
    specs = {
        "gemm_tile[qkv 1536x512]": (NLy, 2.0 * R * 1536 * 512, lambda: K.gemm_tile(
            dt, b["rot"], w["l1.qkv.w"], R, 1536, 512, A2=b["h"], split_n=1024, mode=L.EPI_QKV_HEADS, out=b["Q"],
            out_k=b["K"], out_v=b["V"], scale_q=0.125, Lseq=Lq, Lp=eng.Lp, H=H, n_q=512, n_k=512)),
        "gemm_tile[ffn1 1024x512]": (NLy, 2.0 * R * 1024 * 512, lambda: K.gemm_tile(
            dt, b["h"], w["l1.ff1.w"], R, 1024, 512, bias=w["l1.ff1.b"], act=L.ACT_GELU, out=b["h1"], ldc=1024)),
        "gemm_tile[q 512x512]": (NLy, 2.0 * R * 512 * 512, lambda: K.gemm_tile(
            dt, b["rot"], w["l1.cq.w"], R, 512, 512, mode=L.EPI_QKV_HEADS, out=b["Q"], scale_q=0.125, Lseq=Lq,
            Lp=eng.Lp, H=H, n_q=512, n_k=0)),
        "gemm_rowln[K=512, ln+film+res+ln]": (3 * NLy, 2.0 * R * 512 * 512, lambda: K.gemm_rowln(
            dt, b["O"], w["l1.sfc.w"], R, 512,
            flags=L.ROW_LN_POST | L.ROW_FILM | L.ROW_STORE_X | L.ROW_NEXT_LN | L.ROW_STORE_ROT, ln_g=w["l1.sln.g"],
            ln_b=w["l1.sln.b"], ln_eps=1e-6, film=b["film"], film_ld=fld, xres=b["xa"], xout=b["xa"], Lseq=Lq,
            nln_g=w["l1.norm2.g"], nln_b=w["l1.norm2.b"], nln_eps=1e-5, rout=b["rot"], rope=rope)),
        "gemm_rowln[K=1024, film+res+ln]": (NLy, 2.0 * R * 512 * 1024, lambda: K.gemm_rowln(
            dt, b["h1"], w["l1.ff2.w"], R, 1024, flags=L.ROW_BIAS | L.ROW_FILM | L.ROW_NEXT_LN | L.ROW_STORE_H,
            bias=w["l1.ff2.b"], film=b["film"], film_ld=fld, xres=b["xa"], Lseq=Lq, nln_g=w["l1.norm4.g"],
            nln_b=w["l1.norm4.b"], nln_eps=1e-5, hout=b["h"])),
        "attention[self L=%d]" % Lq: (NLy, 4.0 * 2 * B * H * Lq * Lq * 64, lambda: K.attention(
            dt, b["Q"], b["K"], b["V"], b["O"], 2 * B, H, Lq, Lq, eng.Lp, eng.Lp, 512)),
        "attention[cross M=%d]" % (S + 2): (NLy, 4.0 * 2 * B * H * Lq * (S + 2) * 64, lambda: K.attention(
            dt, b["Q"], b["Kc"][1], b["Vc"][1], b["O"], 2 * B, H, Lq, S + 2, eng.Lp, eng.Lpc, 512, n_shared=B)),
    }
    peak = PEAK_BF16_TFLOPS if dtype == "bf16" else PEAK_F32_TFLOPS
    es = 2 if dtype == "bf16" else 4
    act = lambda cols, e=es: R * cols * e            # one [R, cols] activation matrix
    x32 = R * 512 * 4                                # fp32 residual stream
    qkv_img = 2 * B * H * eng.Lp * 64 * es           # one padded head-major image
    # ALGORITHMIC HBM bytes per launch: every operand read once, every result written once (weights included)
    algo_bytes = {
        "gemm_tile[qkv 1536x512]": 2 * act(512) + 1536 * 512 * es + 3 * act(512),
        "gemm_tile[ffn1 1024x512]": act(512) + 1024 * 512 * es + act(1024),
        "gemm_tile[q 512x512]": act(512) + 512 * 512 * es + act(512),
        "gemm_rowln[K=512, ln+film+res+ln]": act(512) + 512 * 512 * es + x32 + x32 + act(512),
        "gemm_rowln[K=1024, film+res+ln]": act(1024) + 512 * 1024 * es + x32 + act(512),
        "attention[self L=%d]" % Lq: 3 * qkv_img + act(512),
        "attention[cross M=%d]" % (S + 2): qkv_img + 2 * (B + 1) * H * eng.Lpc * 64 * es + act(512),
    }
    rows = {}
    for name, (count, flops, fn) in specs.items():
        ms = event_time_ms(fn)
        nbytes = algo_bytes[name]
        count *= streams
        rows[name] = dict(launches_per_step=count, ms=round(ms, 5), tflops=round(flops / ms / 1e9, 2),
                          frac=round(flops / ms / 1e9 / peak, 4), step_share_ms=round(count * ms, 4),
                          algo_mb=round(nbytes / 1e6, 1), gbps=round(nbytes / ms / 1e6, 1),
                          hbm_frac=round(nbytes / ms / 1e6 / PEAK_HBM_GBPS, 4),
                          flop_per_byte=round(flops / nbytes, 1))
    dom = max(rows, key=lambda k: rows[k]["step_share_ms"])
    d = rows[dom]
    # the roofline that bounds a kernel: HBM when its arithmetic intensity is below the machine balance
    balance = peak * 1e3 / PEAK_HBM_GBPS   # FLOP per byte
    # measured beside this line, from the committed rocprofv3 runs of the same command (profiles/README.md): HBM-side
    # bytes per launch (PMC) and the kernel's average duration INSIDE the sampler, where two free-running streams contend
    traffic, in_step_ms = None, None
    fam = dom.split("[")[0]
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc.json")))
        traffic = pm[fam]["bytes_per_launch"]
    except Exception:
        pass
    try:
        import csv
        sym = {"gemm_rowln": "gemm_rowln_kernel", "gemm_tile": "gemm_tile_kernel", "attention": "attention"}[fam]
        tot = calls = 0
        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r01_kernel_stats_bench_200steps.csv"))):
            if sym in r["Name"]:
                tot += float(r["TotalDurationNs"]); calls += int(r["Calls"])
        in_step_ms = round(tot / calls / 1e6, 5) if calls else None
    except Exception:
        pass
    if d["flop_per_byte"] < balance:
        roof = dict(bound="hbm", kernel=dom, achieved=d["gbps"], peak=PEAK_HBM_GBPS, unit="GB/s", frac=d["hbm_frac"],
                    traffic=traffic, algorithmic_bytes_per_launch=algo_bytes[dom], avg_launch_ms=d["ms"],
                    launches_per_ddpm_step=d["launches_per_step"], mfma_tflops=d["tflops"], mfma_frac=d["frac"],
                    rows_per_launch=R, concurrent_streams=streams, rocprof_in_step_avg_launch_ms=in_step_ms)
    else:
        roof = dict(bound="mfma", kernel=dom, achieved=d["tflops"], peak=peak, unit="TFLOP/s", frac=d["frac"],
                    traffic=traffic, avg_launch_ms=d["ms"], launches_per_ddpm_step=d["launches_per_step"],
                    rows_per_launch=R, concurrent_streams=streams, rocprof_in_step_avg_launch_ms=in_step_ms)
    return roof, rows


def cpu_baseline(dn, S, T, seconds):
    """The CPU oracle (port of the reference's PyTorch path) on the host cores: guided DDPM steps of ONE clip,
    run for ~`seconds`, extrapolated to T steps."""
    from oracle import tcdiff_oracle as O
    sd = O.synth_state_dict(dn=dn, seq_len=S)
    cond = torch.stack([O.synth_cond(0, S)])
    x = torch.stack([O.synth_xT(0, dn * S)])
    tab = O.make_tables(T)
    # pick the intra-op thread count that is fastest on this host (all logical CPUs oversubscribes badly)
    ncpu = os.cpu_count() or 1
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
    n, t0 = 0, time.time()
    with torch.no_grad():
        while True:
            i = T - 1 - n
            x, _ = O.p_sample(sd, tab, x, cond, i, T, 2, torch.randn(x.shape))
            n += 1
            if time.time() - t0 >= seconds or n >= T:
                break
    dt = time.time() - t0
    return dict(value=round(1.0 / (dt / n * T), 6), unit="clips/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{n} of {T} guided DDPM steps of 1 clip ({dn} dancers x {S} frames) on the CPU oracle "
                       f"(torch CPU fp32, {torch.get_num_threads()} threads), {dt:.1f} s, extrapolated x{T}/{n}")


def main():
    a = parse()
    from tcdiff_amd import dist as D
    rank, world, local = D.init_from_env()
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    from tcdiff_amd import DanceDecoder, GaussianDiffusion
    from tcdiff_amd import weights as W

    dn, S, T, B = a.dancers, a.frames, a.ddpm_steps, a.batch
    Lq = dn * S
    model = DanceDecoder(nfeats=151, seq_len=S, latent_dim=512, ff_size=1024, num_layers=8, num_heads=8, dropout=0.1,
                         cond_feature_dim=438, activation=F.gelu, required_dancer_num=dn, compute_dtype=a.dtype)
    model.load_state_dict(W.synth_state_dict_like(model))
    model.eval()
    diff = GaussianDiffusion(model, S, 151, None, schedule="cosine", n_timestep=T, predict_epsilon=False,
                             loss_type="l2", use_p2=False, cond_drop_prob=0.25, guidance_weight=2, seq_len=S)
    diff.to(dev).eval()

    n_total = B * world
    lo, hi = D.shard_range(n_total, rank, world)
    cond = torch.stack([W.synth_cond(c, S) for c in range(lo, hi)]).to(dev)
    xT = torch.stack([W.synth_xT(c, Lq) for c in range(lo, hi)]).to(dev)

    def one_job():
        x = diff.p_sample_loop((hi - lo, Lq, 151), cond, noise=xT, seed=1234, clip_offset=lo)
        return D.gather_samples(x, n_total)

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

    if rank == 0:
        clips_per_s = n_total * a.steps / dt
        gf = GFLOP_PER_CLIP_STEP.get((dn, S))
        # the sampler splits the rank's clips over `streams` sub-batches (tcdiff_amd/diffusion.py); measure those launches
        nb = hi - lo
        streams = diff.dual_parts if diff.dual_stream else 1
        while streams > 1 and nb // streams < 2:
            streams -= 1
        roof, rows = kernel_roofline(model.engine(nb // streams), nb // streams, a.dtype, streams)
        res = {
            "metric": "sampled clips/sec (3 dancers x 150 frames, 1000 DDPM steps)",
            "value": round(clips_per_s, 4), "unit": "clips/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"batch={B} clips/GPU, {dn} dancers x {S} frames, {T} DDPM steps (p_sample_loop, "
                                   f"CFG w=2, cosine schedule), {a.dtype}, {world}xMI355X",
                       "clips_per_gpu": B, "ddpm_steps": T, "tokens_per_clip": Lq},
            "roofline": roof,
        }
        if gf is not None:
            res["whole_path"] = {"algorithmic_gflop_per_clip_step": gf,
                                 "achieved_tflops_per_gpu": round(clips_per_s / world * gf * T / 1e3, 2),
                                 "mfma_frac_per_gpu": round(clips_per_s / world * gf * T / 1e3 /
                                                            (PEAK_BF16_TFLOPS if a.dtype == "bf16" else PEAK_F32_TFLOPS), 4)}
        res["kernels"] = rows
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(dn, S, T, a.cpu_seconds)
        print(json.dumps(res), flush=True)
    D.barrier()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
