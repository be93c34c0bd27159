"""Idle gaps between consecutive kernels of a rocprofv3 --kernel-trace csv (one stream): where a step loses wall time that
is not kernel time.  python tools/gap_analysis.py <kernel_trace.csv> [skip_fraction]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40]) for r in rows))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = ev[int(len(ev) * skip):]                      # steady state: drop warm-up / capture
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
gaps = collections.defaultdict(lambda: [0, 0])
big = []
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    g = max(0, s1 - e0)
    gaps[(n0, n1)][0] += g
    gaps[(n0, n1)][1] += 1
    big.append((g, n0, n1))
print(f"kernels {len(ev)}  span {span / 1e6:.2f} ms  busy {busy / 1e6:.2f} ms  idle {(span - busy) / 1e6:.2f} ms "
      f"({(span - busy) / len(ev) / 1e3:.2f} us per kernel)")
hist = collections.Counter(min(int(g / 1000), 20) for g, _, _ in big)
print("gap histogram (us: count):", sorted(hist.items()))
for (a, b), (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  {t / 1e3:9.1f} us total  {c:5d} x {t / c / 1e3:7.2f} us   after {a}  before {b}")
