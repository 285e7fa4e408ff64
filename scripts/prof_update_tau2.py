"""updateTau per-sweep wall time with the MT19937 (GSL-order) uniforms vs counter-based Philox uniforms: the difference is what
the serial stream still costs after chunked generation."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts, random_state
a = [int(x) for x in sys.argv[1:]]
V, S, G, n = (a + [10000, 64, 8, 200][len(a):])[:4]        # usage: prof_update_tau2.py [V S G sweeps]
counts, tt, gg = synth_counts(V, S, G, 1234)
tau, gamma, eta = random_state(V, S, G, seed=1)
for mode in ("mt", "philox"):
    ctx = _lib.Context(0); ctx.set_counts(counts); ctx.set_state(tau, gamma, eta); ctx.seed(1)
    if mode == "philox": ctx.set_tau_rng(_lib.RNG_PHILOX)
    rng = np.random.default_rng(1)
    gs = np.ascontiguousarray(rng.dirichlet(np.ones(G), size=(n, S)))
    es = np.ascontiguousarray(np.broadcast_to(eta, (n, 4, 4)))
    ctx.update_tau(gs[:10], es[:10])
    t0 = time.perf_counter(); ctx.update_tau(gs, es); dt = time.perf_counter() - t0
    print(mode, "updateTau: %.1f us per sweep" % (1e6 * dt / n))
    ctx.close()
