"""debug: NMFT factorize on a grid of shapes, device vs oracle trace"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
from oracle import cbind, ref_numpy as rn

shapes = [(13000, 40, 3), (13000, 40, 4), (13000, 64, 3), (12000, 40, 3), (13000, 48, 3), (13000, 32, 3), (13000, 16, 3),
          (13000, 40, 8), (14000, 40, 3), (20000, 40, 3), (13000, 33, 5)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for V, S, G in shapes:
    counts, _, _ = synth_counts(V, S, G, seed=1234)
    tau0, gam0 = rn.nmft_random_initialize(np.random.RandomState(1235), V, S, G)
    F = cbind.nmft_freq(counts)
    tc, gc = tau0.copy(), gam0.copy()
    n_ref, tr_ref = cbind.nmft_factorize(F, tc, gc, max_iter=20, min_change=1e-5)
    for fused in (0, 1):
        c = _lib.Context(0); c.set_counts(counts); c.set_nmft_fused(fused); c.nmft_set(tau0, gam0)
        n, tr = c.nmft_factorize(20, 1e-5, False)
        t, g = c.nmft_get()
        m = min(len(tr), len(tr_ref))
        rel = np.abs(tr[:m] - tr_ref[:m]) / np.abs(tr_ref[:m])
        print(V, S, G, "fused", fused, "n", n, n_ref, "max rel trace diff %.2e" % rel.max(), "first bad", int(np.argmax(rel > 1e-9)) if (rel > 1e-9).any() else -1,
              "nan tau", int(np.isnan(t).sum()), "tr tail", tr[-3:], flush=True)
        c.close()
