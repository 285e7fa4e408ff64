"""Times the NMFT update kernels: python scripts/prof_nmft.py [V S G iters]   NMFT_FUSED=0|1 forces the form of the reduce + gamma / control step  (experiment build only -- DESMAN_HIP_LIB=desman_amd/lib/libdesman_hip_ab.so DESMAN_HIP_NMFT_NO_MFMA=1: VALU one-pass kernel)"""
import os, sys, time; sys.path.insert(0, '.')
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
a = sys.argv[1:]
V, S, G = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (10000, 64, 8)
n = int(a[3]) if len(a) > 3 else 100
counts, _, _ = synth_counts(V, S, G, 1234)
ctx = _lib.Context(0); ctx.set_counts(counts)
if os.environ.get("NMFT_FUSED"): ctx.set_nmft_fused(int(os.environ["NMFT_FUSED"]))      # reduce + gamma / control: 0 two launches, 1 one launch, unset: by size
rs = np.random.RandomState(0)
gam0 = np.ascontiguousarray(rs.dirichlet(np.full(G, 0.01), size=S).T)
d = rs.dirichlet(np.full(4, 0.01), size=V * G).reshape(V, G, 4)
tau0 = np.ascontiguousarray(np.transpose(d, (2, 0, 1)).reshape(4 * V, G))
ctx.nmft_set(tau0, gam0); ctx.nmft_factorize(max_iter=5, min_change=0.0)
ctx.nmft_set(tau0, gam0)
t0 = time.perf_counter(); nd, tr = ctx.nmft_factorize(max_iter=n, min_change=0.0); dt = time.perf_counter() - t0
ctx.set_timing(True); ctx.nmft_factorize(max_iter=50, min_change=0.0); tm = ctx.get_timing(); ctx.set_timing(False)
print("V=%d S=%d G=%d: %.1f us per update (wall), kernels" % (V, S, G, 1e6 * dt / nd), {k: round(1e3 * ms / max(c, 1), 1) for k, (ms, c) in tm.items() if c and k.startswith("nmft")}, "div", tr[-1])
