"""profiles/rNN_pmc.json from the two rocprofv3 PMC summaries (tools/pmc_summary.py output of a FETCH_SIZE pass and of
a WRITE_SIZE pass): HBM-side bytes per launch of the hot kernel families.

    python tools/make_pmc_json.py FETCH_summary.txt WRITE_summary.txt out.json [ddpm_steps_profiled | 0 = count sampler_update launches]

With the step count, "bytes_per_ddpm_step" = corrected bytes of EVERY kernel launch of the run / steps.

FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE is doubled (MI355X_MICROARCH.md: gfx950 tallies 128-B read requests at
64 B).  Averages are over ALL launches of the kernel symbol (all epilogue variants, all template instances)."""
import json, re, sys

FAM = {"chain": "chain_kernel", "gemm_rowln": "gemm_rowln_kernel", "gemm_tile": "gemm_tile_kernel", "attention": "attention"}


def parse(path, counter):
    acc = {k: [0.0, 0] for k in FAM}
    for line in open(path):
        m = re.search(r"n=\s*(\d+).*?%s=([0-9.e+]+)" % counter, line)
        if not m:
            continue
        for fam, sym in FAM.items():
            if sym in line.split(" n=")[0]:
                acc[fam][0] += float(m.group(2)) * int(m.group(1))
                acc[fam][1] += int(m.group(1))
    return {k: (v[0] / v[1] if v[1] else None, v[1]) for k, v in acc.items()}


def total_kb(path, counter):
    tot = 0.0
    for line in open(path):
        m = re.search(r"n=\s*(\d+).*?%s=([0-9.e+]+)" % counter, line)
        if m:
            tot += float(m.group(2)) * int(m.group(1))
    return tot


def main():
    f, w = parse(sys.argv[1], "FETCH_SIZE"), parse(sys.argv[2], "WRITE_SIZE")
    out = {"_comment": __doc__.strip().split("\n\n")[-1].replace("\n", " ")}
    for fam in FAM:
        if f[fam][0] is None or w[fam][0] is None:
            continue
        out[fam] = {"launches_profiled": f[fam][1], "fetch_kb_raw": round(f[fam][0], 1), "write_kb": round(w[fam][0], 1),
                    "bytes_per_launch": int((2 * f[fam][0] + w[fam][0]) * 1024)}
    if len(sys.argv) > 4:
        steps = int(sys.argv[4])
        if steps <= 0:                       # count them: one sampler_update launch per DDPM step
            steps = sum(int(re.search(r"n=\s*(\d+)", ln).group(1)) for ln in open(sys.argv[1])
                        if "sampler_update_kernel" in ln)
        out["ddpm_steps_profiled"] = steps
        out["bytes_per_ddpm_step"] = int((2 * total_kb(sys.argv[1], "FETCH_SIZE") + total_kb(sys.argv[2], "WRITE_SIZE"))
                                         * 1024 / steps)
    import importlib.util, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("tcdiff_bench", os.path.join(root, "bench.py"))     # ONE list of kernel sources
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out["source_sha"] = bench.kernel_source_sha()
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out, indent=1))


main()
