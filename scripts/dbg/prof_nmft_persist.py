"""Init_NMFT.factorize per update, persistent loop against the three-launch loop: python scripts/dbg/prof_nmft_persist.py V S G [updates]"""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
a = sys.argv[1:]
V, S, G = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (3000, 96, 8)
n = int(a[3]) if len(a) > 3 else 300
counts, _, _ = synth_counts(V, S, G, 1234)
ctx = _lib.Context(0); ctx.set_counts(counts)
rs = np.random.RandomState(0)
gam0 = np.ascontiguousarray(rs.dirichlet(np.full(G, 0.01), size=S).T)
d = rs.dirichlet(np.full(4, 0.01), size=V * G).reshape(V, G, 4)
tau0 = np.ascontiguousarray(np.transpose(d, (2, 0, 1)).reshape(4 * V, G))
res = {}
for fix in (False, True):
    for persist in (1, 0, 1, 0):
        ctx.set_nmft_persist(persist)
        ctx.nmft_set(tau0, gam0)
        t0 = time.perf_counter(); nd, tr = ctx.nmft_factorize(max_iter=n, min_change=0.0, fix_gamma=fix); dt = time.perf_counter() - t0
        res[(fix, persist)] = (1e6 * dt / nd, tr[-1])
    print("V=%d S=%d G=%d %s: %.1f us per update persistent (where it applies), %.1f three-launch; same objective: %s"
          % (V, S, G, "factorize_tau" if fix else "factorize", res[(fix, 1)][0], res[(fix, 0)][0], res[(fix, 1)][1] == res[(fix, 0)][1]))
