#!/usr/bin/env python3
"""Wall time of a G-sweep (g = 2..8 x 5 seeds = 35 chains, the fan-out of scripts/runDesman.sh) on ONE GPU
for several values of the per-GPU chain concurrency, and with the replicates of a G value batched.  usage: bench_sweep.py [V] [S] [iters]"""
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pandas as p  # noqa: E402

from desman_amd import chains  # noqa: E402
from desman_amd.synth import synth_counts  # noqa: E402

V = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 32
I = int(sys.argv[3]) if len(sys.argv) > 3 else 100
counts, _, _ = synth_counts(V, S, 4, seed=7)
cols = ["Position"] + ["S%d-%s" % (s, b) for s in range(S) for b in "ACGT"]
data = np.concatenate([np.arange(V)[:, None] * 7 + 3, counts.reshape(V, S * 4)], axis=1)
df = p.DataFrame(data, index=["contig%d" % (v // 50) for v in range(V)], columns=cols)
out = {}
with tempfile.TemporaryDirectory() as d:
    freq = os.path.join(d, "syn.freq")
    df.to_csv(freq)
    for c in (1, 2, 4, 8):
        t0 = time.perf_counter()
        sys.stdout = open(os.devnull, "w")
        try:
            chains.main([freq, "--gmin", "2", "--gmax", "8", "--reps", "5", "-i", str(I), "-o", os.path.join(d, "c%d" % c),
                         "-c", str(c)])
        finally:
            sys.stdout = sys.__stdout__
        out["concurrency_%d_wall_s" % c] = time.perf_counter() - t0
    # the replicates of each G as one batched unit (dsm_batch_gibbs_update)
    t0 = time.perf_counter()
    sys.stdout = open(os.devnull, "w")
    try:
        chains.main([freq, "--gmin", "2", "--gmax", "8", "--reps", "5", "-i", str(I), "-o", os.path.join(d, "b5"), "-b", "5", "-c", "1"])
    finally:
        sys.stdout = sys.__stdout__
    out["batch_5_wall_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    sys.stdout = open(os.devnull, "w")
    try:
        chains.main([freq, "--gmin", "2", "--gmax", "8", "--reps", "5", "-i", str(I), "-o", os.path.join(d, "b5c2"), "-b", "5", "-c", "2"])
    finally:
        sys.stdout = sys.__stdout__
    out["batch_5_two_units_at_a_time_wall_s"] = time.perf_counter() - t0
print(json.dumps(dict(V=V, S=S, iters=I, chains=35, **out)))
