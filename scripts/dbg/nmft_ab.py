"""NMFT update time: persistent one-launch path vs the three-launch loop, over shapes"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
shapes = [(10000, 64, 8), (933, 64, 5), (1000, 32, 4), (1000, 16, 5), (3000, 64, 8), (12000, 48, 6), (5000, 32, 3), (200, 16, 4), (10000, 64, 11)]
for V, S, G in shapes:
    counts, _, _ = synth_counts(V, S, G, seed=3)
    rs = np.random.RandomState(0)
    gam0 = np.ascontiguousarray(rs.dirichlet(np.full(G, 0.01), size=S).T)
    d = rs.dirichlet(np.full(4, 0.01), size=V * G).reshape(V, G, 4)
    tau0 = np.ascontiguousarray(np.transpose(d, (2, 0, 1)).reshape(4 * V, G))
    out = []
    for persist in (0, 1):
        c = _lib.Context(0); c.set_counts(counts); c.set_nmft_persist(persist)
        c.nmft_set(tau0, gam0); c.nmft_factorize(20, 0.0)
        c.nmft_set(tau0, gam0)
        n_it = 1000
        t0 = time.perf_counter(); n, tr = c.nmft_factorize(n_it, 0.0); dt = time.perf_counter() - t0
        out.append((1e6 * dt / max(n, 1), n, tr[-1]))
        c.close()
    print("V=%d S=%d G=%d: three-launch %.1f us/update, persistent %.1f us/update (x%.2f), same objective: %s" %
          (V, S, G, out[0][0], out[1][0], out[0][0] / out[1][0], out[0][2] == out[1][2]), flush=True)
