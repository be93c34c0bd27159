"""Diagnostic: VGPR liveness over a straight-line range of a hipcc -S listing (lines a..b of the file): prints the number of
live VGPRs at every `step` lines and the maximum.  python tools/isa_liveness.py file.s first_line last_line [step]"""
import re, sys
path, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
step = int(sys.argv[4]) if len(sys.argv) > 4 else 40
lines = open(path).read().split("\n")[a - 1:b]
NODEST = ("ds_write", "buffer_store", "global_store", "scratch_store", "flat_store", "s_", "v_cmp", ";", "ds_add")
def regs(tok):
    out = []
    for m in re.finditer(r"v\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1): out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.append(int(m.group(3)))
    return out
ins = []
for l in lines:
    s = l.strip()
    if not s or s.startswith((";", ".")) or s.endswith(":"): ins.append(None); continue
    op, _, rest = s.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    if op.startswith(NODEST) or not ops: d, u = [], sum((regs(o) for o in ops), [])
    else:
        d, u = regs(ops[0]), sum((regs(o) for o in ops[1:]), [])
        if op.startswith("v_permlane") or "swap" in op: d = regs(ops[0]) + regs(ops[1]); u = d
    ins.append((d, u))
live, res = set(), []
for i in range(len(ins) - 1, -1, -1):
    if ins[i]:
        d, u = ins[i]
        live -= set(d); live |= set(u)
    res.append((a + i, len(live)))
res.reverse()
mx = max(res, key=lambda t: t[1])
print("max live", mx[1], "at line", mx[0])
for ln, n in res[::step]: print(ln, n)
