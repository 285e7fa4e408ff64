"""Times Init_NMFT.factorize_tau (gamma fixed: the `-r` path, bin/desman:181-206): python scripts/dbg/prof_nmft_tau.py V S G [updates]"""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
a = sys.argv[1:]
V, S, G = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (10000, 64, 8)
n = int(a[3]) if len(a) > 3 else 300
counts, _, gt = synth_counts(V, S, G, 1234)
ctx = _lib.Context(0); ctx.set_counts(counts)
rs = np.random.RandomState(0)
gam0 = np.ascontiguousarray(gt.T)                                   # the abundances a Gibbs run on the selected positions left
d = rs.dirichlet(np.full(4, 0.01), size=V * G).reshape(V, G, 4)
tau0 = np.ascontiguousarray(np.transpose(d, (2, 0, 1)).reshape(4 * V, G))
ctx.nmft_set(tau0, gam0); ctx.nmft_factorize(max_iter=5, min_change=0.0, fix_gamma=True)
res = {}
for persist in (1, 0):
    ctx.set_nmft_persist(persist)
    ctx.nmft_set(tau0, gam0)
    t0 = time.perf_counter(); nd, tr = ctx.nmft_factorize(max_iter=n, min_change=0.0, fix_gamma=True); dt = time.perf_counter() - t0
    res[persist] = (1e6 * dt / nd, tr[-1])
print("V=%d S=%d G=%d factorize_tau: %.1f us per update (persistent where it applies), %.1f (three-launch); div %r %r"
      % (V, S, G, res[1][0], res[0][0], res[1][1], res[0][1]))
