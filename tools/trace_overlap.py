"""Overlap analysis of a rocprofv3 kernel trace of the two-stream sampler: per steady-state window, the time with 0, 1
and 2+ kernels in flight, and the per-queue busy time.  NOTE: under rocprofv3 --kernel-trace the two streams barely
overlap (14 % of the time, 3.4 ms per step instead of 2.1): the tracer serialises dispatches, so this shows the
per-queue kernel time, not the untraced concurrency.  usage: trace_overlap.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"][:40]))
rows.sort()
# a window inside the sampler's replay loop: launches [lo, hi) as fractions of the trace (default 0.15 .. 0.35)
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.15
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 0.35
t0 = rows[int(len(rows) * lo)][0]
t1 = rows[int(len(rows) * hi)][0]
rows = [r for r in rows if r[0] < t1]
ev = []
for s, e, q, n in rows:
    if e <= t0:
        continue
    ev.append((max(s, t0), 1)); ev.append((min(e, t1), -1))
ev.sort()
depth, last, hist = 0, t0, {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    last = t
    depth += d
tot = t1 - t0
print("window %.2f ms, kernels in window %d" % (tot / 1e6, sum(1 for r in rows if r[1] > t0)))
for k in sorted(hist):
    print("  %d kernel(s) in flight: %5.1f %%" % (k, 100.0 * hist[k] / tot))
busy = {}
for s, e, q, n in rows:
    if e > t0:
        busy[q] = busy.get(q, 0) + (min(e, t1) - max(s, t0))
for q, b in sorted(busy.items(), key=lambda x: -x[1])[:6]:
    print("  queue %s busy %5.1f %%" % (q, 100.0 * b / tot))
