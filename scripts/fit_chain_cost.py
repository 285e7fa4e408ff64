#!/usr/bin/env python3
"""Components of a chain's cost on this GPU, per haplotype count: ms per Gibbs iteration and us per NMF update at (V, S) for G = gmin..gmax
(bench.py's own numbers), written as JSON -- what desman_amd/chains.py: chain_cost is fitted on.
usage: fit_chain_cost.py [--V 50000] [--S 96] [--gmin 2] [--gmax 12] [--out gpurun_out/r04/chain_cost_components.json]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--V", type=int, default=50000)
ap.add_argument("--S", type=int, default=96)
ap.add_argument("--gmin", type=int, default=2)
ap.add_argument("--gmax", type=int, default=12)
ap.add_argument("--out", default="gpurun_out/r04/chain_cost_components.json")
a = ap.parse_args()
rows = {}
for G in range(a.gmin, a.gmax + 1):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--V", str(a.V), "--S", str(a.S), "--G", str(G), "--steps", "60",
                        "--warmup", "10", "--repeats", "3", "--no-pmc", "--no-cpu-baseline", "--batch", "0"], capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:                                     # noqa: BLE001
        print("G=%d failed: %s %s" % (G, e, r.stderr[-300:]), flush=True)
        continue
    rows[G] = dict(gibbs_ms_per_iter=d["ms_per_step"], nmft_us_per_update=1e3 * d["nmft"]["ms_per_iter"], stats_spec=d["roofline"]["stats_spec"],
                   kernels_us=d["roofline"]["kernels_us"], nmft_kernels_us=d["nmft"].get("kernels_us"))
    print("G=%2d  gibbs %.4f ms/it  nmft %.1f us/update  %s" % (G, rows[G]["gibbs_ms_per_iter"], rows[G]["nmft_us_per_update"],
                                                              {k: round(v, 1) for k, v in rows[G]["kernels_us"].items()}), flush=True)
os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
json.dump(dict(V=a.V, S=a.S, per_G=rows), open(a.out, "w"), indent=1)
