import json, sys
r = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
T = r["config"]["ddpm_steps"]
print("value=%s %s  ms/ddpm-step=%.3f  => at 1000 steps: %.2f clips/s" % (r["value"], r["unit"], r["ms_per_step"] / T,
      r["config"]["clips_per_gpu"] * r["n_gpus"] / (r["ms_per_step"] / T)))
tot = 0
for k, v in r.get("kernels", {}).items():
    tot += v["step_share_ms"]
    print("%-40s %7.1f us x%2d = %.3f ms  %7.1f TF/s" % (k, v["ms"] * 1000, v["launches_per_step"], v["step_share_ms"], v["tflops"]))
print("sum of listed kernels per step: %.3f ms" % tot)
if "cpu_baseline" in r: print("cpu:", r["cpu_baseline"])
