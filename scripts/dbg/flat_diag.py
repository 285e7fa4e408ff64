#!/usr/bin/env python3
"""Which steps of the tau sweep does the screening pass leave to fp64 in an over-fitted chain?  Runs a chain at G on a table of
--true-G strains, then takes the state and asks the CPU oracle for the four log-probabilities of every step: per haplotype the
abundance range and the histogram of (best - second best) over positions."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from desman_amd import _lib
from desman_amd.synth import synth_counts
from oracle import cbind
ap = argparse.ArgumentParser()
ap.add_argument("--V", type=int, default=5000); ap.add_argument("--S", type=int, default=96)
ap.add_argument("--G", type=int, default=12); ap.add_argument("--true-G", type=int, default=6)
ap.add_argument("--iters", type=int, default=300)
a = ap.parse_args()
V, S, G = a.V, a.S, a.G
counts, _, _ = synth_counts(V, S, a.true_G, seed=1234)
ctx = _lib.Context(0); ctx.set_counts(counts); ctx.seed(0); ctx.set_tau_rng(_lib.RNG_MT19937)
rs = np.random.RandomState(0)
gam0 = np.ascontiguousarray(rs.dirichlet(np.full(G, 0.01), size=S).T)
d = rs.dirichlet(np.full(4, 0.01), size=V * G).reshape(V, G, 4)
ctx.nmft_set(np.ascontiguousarray(np.transpose(d, (2, 0, 1)).reshape(4 * V, G)), gam0)
ctx.nmft_factorize(max_iter=200, min_change=0.0)
_, gam = ctx.nmft_get()
ctx.set_state(ctx.nmft_get_tau(), np.ascontiguousarray(gam.T), 0.96 * np.eye(4) + 0.01)
ctx.gibbs_update(a.iters)
ctx.sweep_stats(reset=True)
ctx.gibbs_update(20)
st, ex = ctx.sweep_stats()
print("G=%d on %d strains, V=%d S=%d: fp64 share of wavefront-steps %.3f" % (G, a.true_G, V, S, ex / max(st, 1)))
tau, gamma, eta = ctx.get_state()
ref = tau.copy()
n, logp = cbind.sample_tau_u(ref, gamma, eta, counts, np.full(V * G, 0.5), want_logp=True)
srt = np.sort(logp, axis=2)
gap = srt[:, :, 3] - srt[:, :, 2]                    # [V, G] best - second
spread = srt[:, :, 3] - srt[:, :, 0]
edges = [0, 0.01, 0.1, 1, 8, 64, 1e30]
print("eta diag", np.round(np.diag(eta), 4))
for g in range(G):
    h = np.histogram(gap[:, g], bins=edges)[0]
    print("g=%2d gamma min %.2e med %.2e max %.2e | gap<0.01 %5d <0.1 %5d <1 %5d <8 %5d <64 %5d >=64 %5d | spread med %.3g" % (
        g, gamma[:, g].min(), np.median(gamma[:, g]), gamma[:, g].max(), *h, np.median(spread[:, g])))
