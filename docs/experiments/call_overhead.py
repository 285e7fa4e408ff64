"""fixed cost of one dsm_ctx_gibbs_update call: wall time of calls of n iterations, n = 1 .. 100 (least-squares intercept and slope)"""
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
ctx.gibbs_update(50)
ns = [1, 2, 5, 10, 20, 50, 100]
res = []
for n in ns:
    ts = []
    for r in range(15):
        t0 = time.perf_counter(); ctx.gibbs_update(n); ts.append(time.perf_counter() - t0)
    res.append(np.median(ts) * 1e6)
    print("n = %3d: %8.1f us per call, %.2f us per iteration" % (n, res[-1], res[-1] / n))
A = np.vstack([np.ones(len(ns)), ns]).T
c, m = np.linalg.lstsq(A, np.array(res), rcond=None)[0]
print("fixed cost per call %.1f us, per iteration %.2f us" % (c, m))
