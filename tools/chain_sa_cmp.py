"""Diagnostic: compare two tools/chain_sa_dump.py files bit for bit (x is column-blocked [64][rows][8])."""
import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
for form in a:
    for k in a[form]:
        x, y = a[form][k].float(), b[form][k].float()
        ne = (x != y) & ~(x.isnan() & y.isnan())
        print(f"{form:16s} {k}: equal={not bool(ne.any())} differing={int(ne.sum())}/{x.numel()} maxdiff={float((x - y).abs().nan_to_num(1e9).max()):.3e} nan={int(x.isnan().sum())}/{int(y.isnan().sum())}")
        if ne.any() and k == "x":
            rows = ne.reshape(64, -1, 8).any(dim=2).any(dim=0).nonzero().flatten().tolist()
            R = ne.reshape(64, -1, 8).shape[1]
            print(f"   differing rows: {len(rows)} of {R}:", rows[:40], "..." if len(rows) > 40 else "")
            nanrows = y.isnan().reshape(64, -1, 8).any(dim=2).any(dim=0).nonzero().flatten().tolist()
            if nanrows: print("   NaN rows in the second file:", nanrows[:40])
