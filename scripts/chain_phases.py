#!/usr/bin/env python3
"""Where a whole `desman` chain spends its wall time, per haplotype count (host + GPU phases by cProfile cumulative times): the inputs of
the scheduler's cost model (desman_amd/chains.py: chain_cost).  usage: chain_phases.py [--V 50000] [--S 96] [-i 500] [--gs 2,3,4,6,8,10,12]"""
import argparse
import cProfile
import json
import os
import pstats
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pandas as p  # noqa: E402

from desman_amd import cli  # noqa: E402
from desman_amd.synth import synth_counts  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--V", type=int, default=50000)
ap.add_argument("--S", type=int, default=96)
ap.add_argument("-i", "--iters", type=int, default=500)
ap.add_argument("--gs", default="2,3,4,6,8,10,12")
ap.add_argument("--out", default="gpurun_out/r04/chain_phases.json")
a = ap.parse_args()
V, S = a.V, a.S
counts, _, _ = synth_counts(V, S, 6, seed=1234)
cols = ["Position"] + ["S%d-%s" % (s, b) for s in range(S) for b in "ACGT"]
data = np.concatenate([np.arange(V)[:, None] * 7 + 3, counts.reshape(V, S * 4)], axis=1)
df = p.DataFrame(data, index=["contig%d" % (v // 50) for v in range(V)], columns=cols)
KEYS = {"nmft_factorize": ("Init_NMFT.py", "factorize"), "nmft_init_draws": ("Init_NMFT.py", "random_initialize"),
        "gibbs_update": ("HaploSNP_Sampler.py", "update"), "remove_degenerate": ("HaploSNP_Sampler.py", "removeDegenerate"),
        "sampler_ctor": ("HaploSNP_Sampler.py", "__init__"), "read_table": ("cli.py", "_read_table"),
        "variant_filter": ("Variant_Filter.py", "__init__")}
rows = {}
with tempfile.TemporaryDirectory() as d:
    freq = os.path.join(d, "syn.freq")
    df.to_csv(freq)
    cli.main([freq, "-g", "3", "-i", "5", "-o", os.path.join(d, "warm")])       # warm: library load, table cache
    for G in [int(x) for x in a.gs.split(",")]:
        import logging
        for h in logging.root.handlers[:]:                      # one log file per chain (basicConfig configures only an empty root logger)
            logging.root.removeHandler(h)
        pr = cProfile.Profile()
        t0 = time.perf_counter()
        pr.enable()
        cli.main([freq, "-g", str(G), "-i", str(a.iters), "-o", os.path.join(d, "out%d" % G)])
        pr.disable()
        wall = time.perf_counter() - t0
        st = pstats.Stats(pr)
        ph = {}
        out_s = 0.0
        for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
            base = os.path.basename(fn)
            for k, (f, n) in KEYS.items():
                if base == f and name == n:
                    ph[k] = ph.get(k, 0.0) + ct
            if base == "Output_Results.py" and name.startswith("output") or (base == "Output_Results.py" and name == "outputOptimalFit"):
                out_s += ct
        ph["output_files"] = out_s
        ph["wall"] = wall
        ph["other"] = wall - sum(v for k, v in ph.items() if k not in ("wall", "nmft_init_draws"))
        try:
            log = open(os.path.join(d, "out%d" % G, "log_file.txt")).read()
            ph["nmft_updates_logged"] = max([int(ln.split("NTF Iter ")[1].split(",")[0]) for ln in log.splitlines() if "NTF Iter " in ln] or [0])
        except OSError:
            ph["nmft_updates_logged"] = None
        rows[G] = ph
        print("G=%2d wall %.2f s  " % (G, wall) + "  ".join("%s %s" % (k, ("%.2f" % v) if isinstance(v, float) else v) for k, v in ph.items() if k != "wall"), flush=True)
os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
json.dump(dict(V=V, S=S, iters=a.iters, per_G=rows), open(a.out, "w"), indent=1)
