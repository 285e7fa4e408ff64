"""Dumps the chain state the benchmark times (NMFT init + n Gibbs iterations) for offline analysis."""
import sys; sys.path.insert(0, '.')
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
V, S, G = 10000, 64, 8
n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
counts, tt, gg = synth_counts(V, S, G, seed=1234)
ctx = _lib.Context(0); ctx.set_counts(counts); ctx.seed(0)
rs = np.random.RandomState(0)
gam0 = np.ascontiguousarray(rs.dirichlet(np.full(G, 0.01), size=S).T)
d = rs.dirichlet(np.full(4, 0.01), size=V * G).reshape(V, G, 4)
tau0 = np.ascontiguousarray(np.transpose(d, (2, 0, 1)).reshape(4 * V, G))
ctx.nmft_set(tau0, gam0)
ctx.nmft_factorize(max_iter=200, min_change=0.0)
tau_init = ctx.nmft_get_tau(); _, gam = ctx.nmft_get()
ctx.set_state(tau_init, np.ascontiguousarray(gam.T), 0.96 * np.eye(4) + 0.01)
out = {}
for k in (0, 5, n, 100, 500):
    done = sum(x for x in out.get("_done", [0]))
    ctx.gibbs_update(k - done if k > done else 0)
    out.setdefault("_done", []).append(k - done if k > done else 0)
    t, g, e = ctx.get_state()
    out["tau_%d" % k] = np.argmax(t, axis=2).astype(np.uint8); out["gamma_%d" % k] = g; out["eta_%d" % k] = e
    ll, lp = ctx.loglik()
    print(k, "lp", lp, "mismatch vs truth (unpermuted)", float((out["tau_%d" % k] != tt).mean()))
del out["_done"]
np.savez_compressed("gpurun_out/bench_state.npz", **out)
