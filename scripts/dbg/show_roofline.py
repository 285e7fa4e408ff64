"""one line of a bench.py JSON line read from stdin: the roofline's durations (event bracket, dispatch latency, the kernel's own)"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d["roofline"]
print("ms/step %.4f" % d["ms_per_step"], "frac %.4f" % r["frac"], "avg %.1f" % r["avg_kernel_us"], "bracket %.1f" % r["avg_event_bracket_us"],
      "dispatch %.2f" % r["dispatch_latency_us_per_launch"],
      {k: (round(v["avg_kernel_us"], 1), round(v.get("shader_clock_GHz_pmc", 0), 2)) for k, v in r["per_kernel"].items()},
      {k: round(v, 1) for k, v in r["kernels_us"].items()})
