import json, sys
line = [l for l in open(sys.argv[1]).read().strip().split("\n") if l.startswith("{")][-1]
r = json.loads(line)
T = r["config"]["ddpm_steps"]
print("value=%s %s  ms/ddpm-step=%.4f" % (r["value"], r["unit"], r["ms_per_step"] / T))
for k, v in r.get("kernels", {}).items():
    print("%-12s %6.1f launches/step  avg %7.2f us  %.3f ms/step  share %.3f  %7.1f TF/s (%.3f of peak)" % (
        k, v["launches_per_ddpm_step"], v["avg_launch_ms"] * 1e3, v["device_ms_per_ddpm_step"], v["share_of_gpu_time"],
        v["tflops"], v["mfma_frac"]))
if r.get("roofline"): print("roofline:", {k: r["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms")})
for k in ("parity_mode", "cpu_baseline", "kernel_profile_error"):
    if k in r: print(k + ":", r[k])
