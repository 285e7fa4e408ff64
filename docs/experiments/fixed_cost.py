import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
from oracle import cbind
V, S, G = 10000, 64, 8
counts, tt, gg = synth_counts(V, S, G, 1234)
ctx = _lib.Context(0); ctx.set_counts(counts); ctx.seed(0)
ctx.set_state(cbind.idx_to_onehot(tt), np.ascontiguousarray(gg), 0.96 * np.eye(4) + 0.01)
ctx.gibbs_update(100)
for n in (1, 2, 5, 10, 20, 50, 100, 500):
    ts = []
    for r in range(5):
        t0 = time.perf_counter(); ctx.gibbs_update(n); ts.append(time.perf_counter() - t0)
    print("n=%d: %.1f us per call, %.1f us per step (min of 5)" % (n, 1e6 * min(ts), 1e6 * min(ts) / n), flush=True)
