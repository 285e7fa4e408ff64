"""Cost of one legacy-shim call (module swap use): sampletau.sample_tau on V=10k, S=64, G=8."""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from desman_amd import sampletau
from desman_amd.synth import synth_counts, random_state
V, S, G = 10000, 64, 8
counts, _, _ = synth_counts(V, S, G, 1234)
tau, gamma, eta = random_state(V, S, G, seed=1)
sampletau.initRNG(); sampletau.setRNG(1)
sampletau.sample_tau(tau, gamma, eta, counts)
t0 = time.perf_counter()
for _ in range(20):
    sampletau.sample_tau(tau, gamma, eta, counts)
print("legacy sample_tau: %.2f ms per call (upload of the %.0f MB tensor + sweep + download)" % (50 * (time.perf_counter() - t0), counts.nbytes / 1e6))
sampletau.freeRNG()
