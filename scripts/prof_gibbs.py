"""The workload of the PMC passes (scripts/collect_profiles.sh, bench.py --pmc): the benchmark's chain (NMFT-initialised) stepping n
Gibbs iterations -- the same state the timed region of bench.py runs in.  usage: prof_gibbs.py [n] [V S G [depth_scale]]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
V, S, G = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (10000, 64, 8)
depth = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
counts, tt, gg = synth_counts(V, S, G, 1234, depth_scale=depth)
ctx = _lib.Context(0); ctx.set_counts(counts); ctx.seed(0)
rs = np.random.RandomState(0)
gam0 = np.ascontiguousarray(rs.dirichlet(np.full(G, 0.01), size=S).T) if G > 1 else np.ones((G, S))
d = rs.dirichlet(np.full(4, 0.01), size=V * G).reshape(V, G, 4)
tau0 = np.ascontiguousarray(np.transpose(d, (2, 0, 1)).reshape(4 * V, G))
ctx.nmft_set(tau0, gam0)
ctx.nmft_factorize(max_iter=200, min_change=0.0)
tau_init = ctx.nmft_get_tau(); _, gam = ctx.nmft_get()
ctx.set_state(tau_init, np.ascontiguousarray(gam.T), 0.96 * np.eye(4) + 0.01)
ctx.gibbs_update(n)
print(ctx.get_trace()["ll"][-1])
