#!/usr/bin/env python3
"""What a Gibbs iteration costs when the chain fits the table with too few or too many haplotypes -- the normal case in a G-sweep
(BASELINE config 5: g = 2..12 on one table).  bench.py at (V, S) for every G, the table generated from --true-G strains; per G:
ms per iteration, per-kernel us, share of the tau sweep's wavefront-steps left to the fp64 code, mu/E specification.
usage: misfit_scan.py [--V 50000] [--S 96] [--true-G 6] [--gs 2,3,4,6,8,10,12] [--out gpurun_out/r04/misfit_scan.json]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--V", type=int, default=50000)
ap.add_argument("--S", type=int, default=96)
ap.add_argument("--true-G", type=int, default=6)
ap.add_argument("--gs", default="2,3,4,6,8,10,12")
ap.add_argument("--warmup", type=int, default=300)
ap.add_argument("--stats-spec", type=int, default=0)
ap.add_argument("--out", default="gpurun_out/r04/misfit_scan.json")
a = ap.parse_args()
rows = {}
for G in [int(x) for x in a.gs.split(",")]:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--V", str(a.V), "--S", str(a.S), "--G", str(G), "--true-G", str(a.true_G),
                        "--stats-spec", str(a.stats_spec), "--steps", "60", "--warmup", str(a.warmup), "--repeats", "3", "--no-pmc", "--no-cpu-baseline", "--batch", "0"],
                       capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:                                     # noqa: BLE001
        print("G=%d failed: %s %s" % (G, e, r.stderr[-300:]), flush=True)
        continue
    rf = d["roofline"]
    rows[G] = dict(ms_per_iter=d["ms_per_step"], kernels_us=rf["kernels_us"], stats_spec=rf["stats_spec"],
                   tau_steps_fp64_frac=rf.get("tau_steps_fp64_frac"))
    print("G=%2d (table from %d strains)  %.4f ms/it  fp64 steps %.3f  spec %s  %s" % (
        G, a.true_G, d["ms_per_step"], rf.get("tau_steps_fp64_frac") or 0.0, rf["stats_spec"],
        {k: round(v, 1) for k, v in rf["kernels_us"].items()}), flush=True)
os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
json.dump(dict(V=a.V, S=a.S, true_G=a.true_G, per_G=rows), open(a.out, "w"), indent=1)
