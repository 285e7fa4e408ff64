"""The workload of the PMC passes (scripts/collect_profiles.sh): the benchmark's chain (NMFT-initialised, configs[2]
shape) stepping n Gibbs iterations -- the same state the timed region of bench.py runs in."""
import sys; sys.path.insert(0, '.')
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
V, S, G = 10000, 64, 8
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
counts, tt, gg = synth_counts(V, S, G, 1234)
ctx = _lib.Context(0); ctx.set_counts(counts); ctx.seed(0)
rs = np.random.RandomState(0)
gam0 = np.ascontiguousarray(rs.dirichlet(np.full(G, 0.01), size=S).T)
d = rs.dirichlet(np.full(4, 0.01), size=V * G).reshape(V, G, 4)
tau0 = np.ascontiguousarray(np.transpose(d, (2, 0, 1)).reshape(4 * V, G))
ctx.nmft_set(tau0, gam0)
ctx.nmft_factorize(max_iter=200, min_change=0.0)
tau_init = ctx.nmft_get_tau(); _, gam = ctx.nmft_get()
ctx.set_state(tau_init, np.ascontiguousarray(gam.T), 0.96 * np.eye(4) + 0.01)
ctx.gibbs_update(n)
print(ctx.get_trace()["ll"][-1])
