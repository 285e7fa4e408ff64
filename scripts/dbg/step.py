import faulthandler, sys, os
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts, random_state
print("load", flush=True)
c = _lib.Context(0); print("ctx", flush=True)
V, S, G = 2000, 64, 8
counts, _, _ = synth_counts(V, S, G, seed=1)
tau, gamma, eta = random_state(V, S, G, seed=2)
c.set_counts(counts); print("counts", flush=True)
c.set_state(tau, gamma, eta); print("state", flush=True)
c.seed(1, ctr_seed=5)
print("launch info", c.tau_launch_info(), flush=True)
for spec in (1, 2, 3):
    c.force_stats_spec(spec); print("spec", c.stats_spec(), flush=True)
    mu, E = c.sample_stats(0); print("stats", mu.sum(), flush=True)
c.force_stats_spec(0)
c.gibbs_update(3); print("gibbs", flush=True)
