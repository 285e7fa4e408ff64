"""ms per Gibbs iteration and per-kernel us over a grid of shapes (looks for cliffs): python scripts/shape_scan.py"""
import json, subprocess, sys
rows = []
for V in (2000, 20000, 100000):
    for S in (8, 32, 64, 128, 300):
        for G in (2, 5, 8, 12, 16):
            if V * S > 100000 * 128: continue
            steps = 40 if V * S > 2e6 else 100
            out = subprocess.run([sys.executable, "bench.py", "--V", str(V), "--S", str(S), "--G", str(G), "--steps", str(steps),
                                  "--warmup", "10", "--no-cpu-baseline", "--batch", "0"], capture_output=True, text=True).stdout.strip().split("\n")[-1]
            try:
                d = json.loads(out)
                k = d["roofline"]["kernels_us"]
                print("V=%6d S=%3d G=%2d  %.4f ms  %5.2f ns per V*S  spec %s  %s" % (V, S, G, d["ms_per_step"], 1e6 * d["ms_per_step"] / (V * S),
                      d["roofline"].get("stats_spec"), {a: round(b, 1) for a, b in k.items() if a != "mt"}), flush=True)
            except Exception as e:
                print("V=%d S=%d G=%d failed: %s %s" % (V, S, G, e, out[:200]), flush=True)
