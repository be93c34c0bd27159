#!/bin/bash
# long same-box A/B: round-3 tree against the working tree, N alternating pairs of `bench.py --steps 4` (4 jobs of 16 clips x 1000 steps)
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:?run under gpurun (or export GRAFT_REPO_ROOT=the repo root)}"
B="--steps 4 --warmup 1 --no-kernel-profile --no-parity-mode --no-cpu-baseline --no-train-step --no-other-configs"
for rep in $(seq 1 ${1:-8}); do
  (cd tools/probe/r3_tree && python bench.py $B 2>/dev/null | python ../../show_bench.py /dev/stdin | sed 's/^/r3:  /')
  python bench.py $B 2>/dev/null | python tools/show_bench.py /dev/stdin | sed 's/^/r4:  /'
done | tee gpurun_out/r04_ab_vs_r3.txt
python - <<'PY'
import re
a = {"r3": [], "r4": []}
for l in open("gpurun_out/r04_ab_vs_r3.txt"):
    m = re.match(r"(r\d):\s+value=([0-9.]+)", l)
    if m: a[m.group(1)].append(float(m.group(2)))
for k, v in a.items():
    v = sorted(v)
    print(k, "n=%d mean %.3f median %.3f min %.3f max %.3f" % (len(v), sum(v) / len(v), v[len(v) // 2], v[0], v[-1]))
PY
timeout 120 python tools/power_watch.py 10 > gpurun_out/r04_power_watch.txt 2>&1; head -3 gpurun_out/r04_power_watch.txt; grep "^run" gpurun_out/r04_power_watch.txt | sed -n '5,12p'
