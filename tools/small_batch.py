"""Small-batch latency of the samplers: config 1 (1 clip, 2 x 60, 100 DDPM steps) and what TCDiff.py calls when
it renders (ddim_sample, 50 steps, 1 / 4 clips of 3 x 150) under the three launch modes of the decoder layer: TCDIFF_CHAIN=2
fused layer chain (one 64-row block per CU: 4 blocks for a 2 x 120-row job), =1 chain A + B, =0 op-by-op tiles.
Each mode runs in its own process (the switch is read when the engine is built)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import json, os, sys, time, types
sys.path.insert(0, os.environ["TC_ROOT"])
import torch
import bench
a = types.SimpleNamespace(dtype="bf16")
out = bench.other_configs(a, torch.device("cuda", 0))
print("RESULT " + json.dumps({k: [v["value"], v["ms_per_step"]] for k, v in out.items() if "config4" not in k}))
'''
# "2s": the fused mode with the small-job form of the layer switched off (TCDIFF_SPLIT=0: one workgroup per 16-row block, round 5);
# "2m" / "2n": the small-job form with parts 1 + 2 always / never merged (TCDIFF_SPLIT_MERGE)
for mode in sys.argv[1:] or ["2", "2s", "1", "0"]:
    env = dict(os.environ, TC_ROOT=ROOT, TCDIFF_CHAIN=mode[0], TCDIFF_SPLIT="0" if mode.endswith("s") else "1")
    if mode.endswith("m") or mode.endswith("n"):
        env["TCDIFF_SPLIT_MERGE"] = "1" if mode.endswith("m") else "0"
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=1200)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    print(f"TCDIFF_CHAIN={mode[0]} TCDIFF_SPLIT={env['TCDIFF_SPLIT']} TCDIFF_SPLIT_MERGE={env.get('TCDIFF_SPLIT_MERGE', 'default')}:",
          line[-1][7:] if line else r.stderr[-800:])
