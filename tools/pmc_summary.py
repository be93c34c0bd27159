"""Summarise a rocprofv3 --pmc run: per kernel name, mean counter value per dispatch and mean duration."""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
cnt = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "?").split("(")[0][:60]
        cnt[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "?").split("(")[0][:60]
        dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
names = sorted(cnt, key=lambda k: -sum(dur.get(k, [0])))
for k in names:
    c = cnt[k]
    line = f"{k:60s} n={len(next(iter(c.values()))):6d} us={sum(dur[k])/max(1,len(dur[k])):9.2f} "
    line += " ".join(f"{n}={sum(v)/len(v):.4g}" for n, v in sorted(c.items()))
    print(line)
