"""A real chain's tau sweep, phase by phase: the NMF start (5000 updates), the burn-in, removeDegenerate, the sampling phase -- share of
wavefront-steps left to fp64 per phase, and at the end the per-haplotype abundance range and the gaps (best - second best candidate
log-probability) over the positions, by the CPU oracle.  usage: chain_fp64.py G [V S true_G iters]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from numpy.random import RandomState
from desman_amd import sampletau
from desman_amd.Init_NMFT import Init_NMFT
from desman_amd.HaploSNP_Sampler import HaploSNP_Sampler
from desman_amd.synth import synth_counts
from oracle import cbind
G = int(sys.argv[1]); V = int(sys.argv[2]) if len(sys.argv) > 2 else 50000; S = int(sys.argv[3]) if len(sys.argv) > 3 else 96
tG = int(sys.argv[4]) if len(sys.argv) > 4 else 6; it = int(sys.argv[5]) if len(sys.argv) > 5 else 500
counts, _, _ = synth_counts(V, S, tG, seed=1234)
rng = RandomState(0)
sampletau.initRNG(); sampletau.setRNG(0)
nm = Init_NMFT(counts, G, rng)
t0 = time.perf_counter(); nm.factorize(); print("NMF start %.2f s" % (time.perf_counter() - t0))
ch = HaploSNP_Sampler(counts, G, rng, max_iter=it, ctx=nm._ctx)
ch.tau = np.copy(nm.get_tau(), order='C'); ch.updateTauIndices(); ch.gamma = np.copy(nm.get_gamma(), order='C')
ch.eta = 0.96 * np.eye(4) + 0.01


def report(tag):
    st, ex = ch._ctx.sweep_stats(reset=True)
    print("%s: fp64 share of wavefront-steps %.3f" % (tag, ex / max(st, 1)))
    gamma = ch.gamma
    ref = ch.tau.copy()
    n, logp = cbind.sample_tau_u(ref, np.ascontiguousarray(gamma), np.ascontiguousarray(ch.eta), counts, np.full(V * ch.G, 0.5), want_logp=True)
    srt = np.sort(logp, axis=2)
    gap = srt[:, :, 3] - srt[:, :, 2]
    edges = [0, 0.01, 0.1, 1, 4, 8, 64, 1e30]
    for g in range(ch.G):
        h = np.histogram(gap[:, g], bins=edges)[0]
        print("  g=%2d gamma min %.1e med %.1e max %.1e | gap<0.01 %5d <0.1 %5d <1 %5d <4 %5d <8 %5d <64 %5d >=64 %5d" % (
            g, gamma[:, g].min(), np.median(gamma[:, g]), gamma[:, g].max(), *h))


ch._ctx.sweep_stats(reset=True)
t0 = time.perf_counter(); ch.update(); print("burn-in %.2f s" % (time.perf_counter() - t0)); report("after burn-in")
ch.removeDegenerate(); print("haplotypes kept", ch.G)
t0 = time.perf_counter(); ch.update(); print("sampling %.2f s" % (time.perf_counter() - t0)); report("after sampling")
