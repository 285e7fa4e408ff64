"""us per NMFT update over a grid of shapes (looks for cliffs): python scripts/shape_scan_nmft.py [quick]
Prints wall us per update of factorize() (the path the product takes for the shape), the bytes of F it passes over per update and
the rate that makes, so that a shape that falls off its neighbours stands out."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
quick = len(sys.argv) > 1
for V in ((20000,) if quick else (2000, 10000, 50000, 200000)):
    for S in (8, 16, 32, 48, 64, 80, 96, 128, 300):
        if V * S > 200000 * 128: continue
        counts, _, _ = synth_counts(V, S, 4, 1234)
        ctx = _lib.Context(0); ctx.set_counts(counts)
        for G in (2, 5, 8, 12, 16):
            rs = np.random.RandomState(0)
            gam0 = np.ascontiguousarray(rs.dirichlet(np.full(G, 0.01), size=S).T)
            d = rs.dirichlet(np.full(4, 0.01), size=V * G).reshape(V, G, 4)
            tau0 = np.ascontiguousarray(np.transpose(d, (2, 0, 1)).reshape(4 * V, G))
            n = 200 if V * S <= 2e6 else 60
            try:
                ctx.nmft_set(tau0, gam0); ctx.nmft_factorize(max_iter=5, min_change=0.0)
                ctx.nmft_set(tau0, gam0)
                t0 = time.perf_counter(); nd, tr = ctx.nmft_factorize(max_iter=n, min_change=0.0); dt = time.perf_counter() - t0
                us = 1e6 * dt / nd
                print("V=%6d S=%3d G=%2d  %8.1f us/update  F %7.1f MB  %6.0f GB/s  %.3f ns per F element" % (V, S, G, us, V * 4 * S * 8 / 1e6, V * 4 * S * 8 / us / 1e3, 1e3 * us / (V * 4 * S)), flush=True)
            except Exception as e:
                print("V=%d S=%d G=%d failed: %s" % (V, S, G, e), flush=True)
        del ctx
