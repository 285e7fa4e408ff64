#!/usr/bin/env python
"""Wall-clock of the whole `desman` program (CSV in, result files out) on a synthetic base-count table of the
bench shape, with the per-stage split: python scripts/bench_cli.py [--V 10000 --S 64 --G 8 --iters 500]."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--V", type=int, default=10000)
    ap.add_argument("--S", type=int, default=64)
    ap.add_argument("--G", type=int, default=8)
    ap.add_argument("--iters", type=int, default=500)
    a = ap.parse_args()
    from desman_amd.synth import synth_counts
    from desman_amd import cli
    counts, _, _ = synth_counts(a.V, a.S, a.G, seed=11)
    with tempfile.TemporaryDirectory() as td:
        cols = ["S%d-%s" % (s, b) for s in range(a.S) for b in "ACGT"]
        df = pd.DataFrame(counts.reshape(a.V, a.S * 4), columns=cols, index=["c%d" % (v // 50) for v in range(a.V)])
        df.insert(0, "Position", np.arange(a.V) % 50 * 7)
        df.index.name = "Contig"
        path = os.path.join(td, "syn.freq")
        df.to_csv(path)
        stages = {}
        marks = [("load", cli._load), ("fit", cli._fit), ("report", cli._report)]
        for name, fn in marks:
            def wrap(*args, _fn=fn, _name=name, **kw):
                t0 = time.perf_counter()
                r = _fn(*args, **kw)
                stages[_name] = time.perf_counter() - t0
                return r
            setattr(cli, fn.__name__, wrap)
        t0 = time.perf_counter()
        cli.main([path, "-g", str(a.G), "-i", str(a.iters), "-o", os.path.join(td, "out"), "-s", "3"])
        total = time.perf_counter() - t0
        fit = open(os.path.join(td, "out", "fit.txt")).read().strip()
    print(json.dumps({"workload": "desman syn.freq -g %d -i %d (V=%d, S=%d): CSV -> result files" % (a.G, a.iters, a.V, a.S),
                      "wall_s": total, "stages_s": stages, "gibbs_iterations": 2 * a.iters, "fit": fit}))


if __name__ == "__main__":
    main()
