"""Diagnostic: VGPRs / scratch / spills per kernel from a `hipcc -Rpass-analysis=kernel-resource-usage` log.
python tools/res_usage.py log.txt [filter]"""
import re, sys
cur, rows = None, {}
for l in open(sys.argv[1]):
    m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|TotalSGPRs): (\S+)", l)
    if not m: continue
    k, v = m.groups()
    if k == "Function Name": cur = v; rows[cur] = {}
    elif cur: rows[cur][k.split()[0] + (" Spill" if "Spill" in k else "")] = v
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f, r in rows.items():
    if flt in f: print(f"{f:60s} VGPR {r.get('VGPRs'):>4s} AGPR {r.get('AGPRs'):>3s} scratch {r.get('ScratchSize'):>5s} vspill {r.get('VGPRs Spill'):>4s} sspill {r.get('SGPRs Spill'):>4s}")
