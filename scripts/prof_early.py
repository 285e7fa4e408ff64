"""Per-kernel time of the first Gibbs iterations after the NMFT initialisation (the window the driver's bench times)."""
import sys; sys.path.insert(0, '.')
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
V, S, G = 10000, 64, 8
nm = int(sys.argv[1]) if len(sys.argv) > 1 else 200
counts, tt, gg = synth_counts(V, S, G, seed=1234)
ctx = _lib.Context(0); ctx.set_counts(counts); ctx.seed(0)
rs = np.random.RandomState(0)
gam0 = np.ascontiguousarray(rs.dirichlet(np.full(G, 0.01), size=S).T)
d = rs.dirichlet(np.full(4, 0.01), size=V * G).reshape(V, G, 4)
tau0 = np.ascontiguousarray(np.transpose(d, (2, 0, 1)).reshape(4 * V, G))
ctx.nmft_set(tau0, gam0)
ctx.nmft_factorize(max_iter=nm, min_change=0.0)
tau_init = ctx.nmft_get_tau(); _, gam = ctx.nmft_get()
ctx.set_state(tau_init, np.ascontiguousarray(gam.T), 0.96 * np.eye(4) + 0.01)
done = 0
for n in (1, 1, 1, 2, 5, 5, 10, 25, 50):
    ctx.set_timing(True); ctx.gibbs_update(n); tm = ctx.get_timing(); ctx.set_timing(False)
    done += n
    print("iterations %3d-%3d:" % (done - n, done), {k: round(1e3 * ms / max(c, 1), 1) for k, (ms, c) in tm.items() if c and k in ("stats", "dirichlet", "tau")},
          "nchange", int(ctx.get_trace()["nchange"][-1]))
import time
for pause in (0.0, 2.0, 0.0):
    time.sleep(pause)
    ctx.set_timing(True); ctx.gibbs_update(5); tm = ctx.get_timing(); ctx.set_timing(False)
    print("after %.0f s idle, 5 iterations:" % pause, {k: round(1e3 * ms / max(c, 1), 1) for k, (ms, c) in tm.items() if c and k in ("stats", "dirichlet", "tau")})
