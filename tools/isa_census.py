"""Diagnostic: static instruction census of one kernel in a hipcc -S listing, per basic block (label to label):
MFMA / transcendental / other VALU / LDS / vector memory / waits / barriers.  python tools/isa_census.py file.s SYMBOL"""
import re, sys
path, sym = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
TRANS = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")
blocks, cur = [], {"name": "entry", "n": {}}
def bump(k): cur["n"][k] = cur["n"].get(k, 0) + 1
for l in lines[start + 1:end + 1]:
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", s):
            blocks.append(cur); cur = {"name": s.split(":")[0], "n": {}}
        continue
    op = s.split()[0]
    if op.startswith("v_mfma"): bump("mfma")
    elif op.startswith(TRANS): bump("trans")
    elif op.startswith("v_pk_"): bump("valu_pk")
    elif op.startswith("v_"): bump("valu")
    elif op.startswith("ds_"): bump("lds")
    elif op.startswith(("buffer_", "global_", "flat_", "scratch_")): bump("vmem_st" if "store" in op else "vmem_ld")
    elif op == "s_waitcnt": bump("waitcnt")
    elif op == "s_barrier": bump("barrier")
    elif op.startswith("s_cbranch") or op == "s_branch": cur["n"]["br"] = cur["n"].get("br", "") + " " + s.split()[-1]
    elif op.startswith("s_"): bump("salu")
blocks.append(cur)
keys = ["mfma", "trans", "valu", "valu_pk", "lds", "vmem_ld", "vmem_st", "waitcnt", "barrier", "salu"]
print(f"{'block':12s} " + " ".join(f"{k:>8s}" for k in keys) + "  branches")
tot = {k: 0 for k in keys}
for b in blocks:
    if not b["n"]: continue
    print(f"{b['name']:12s} " + " ".join(f"{b['n'].get(k, 0):8d}" for k in keys) + "  " + str(b["n"].get("br", "")))
    for k in keys: tot[k] += b["n"].get(k, 0)
print(f"{'TOTAL(static)':12s} " + " ".join(f"{tot[k]:8d}" for k in keys))
