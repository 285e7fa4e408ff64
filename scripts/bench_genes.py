#!/usr/bin/env python
"""Throughput of the accessory-gene sampler on one GPU: batched (rng='philox') iterations of
Eta_Sampler.update over C genes, against the CPU oracle's reference-order loop on a bounded sample of the genes.

    python scripts/bench_genes.py [--genes 2000 --samples 32 --haplotypes 6 --vmax 20 --iters 50]
prints one JSON line: gene-haplotype copy-number updates per second.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genes", type=int, default=2000)
    ap.add_argument("--samples", type=int, default=32)
    ap.add_argument("--haplotypes", type=int, default=6)
    ap.add_argument("--vmax", type=int, default=20)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--cpu-genes", type=int, default=40, help="genes of the CPU oracle sample (0 = skip)")
    args = ap.parse_args()
    from scipy.special import gammaln
    from desman_amd import _lib
    from desman_amd.synth import synth_genes
    C, S, G = args.genes, args.samples, args.haplotypes
    d = synth_genes(C, S, G, seed=5, vmax=args.vmax, mean_lo=2.0, mean_hi=20.0)
    off = np.concatenate([[0], np.cumsum(np.bincount(d['gene_of'], minlength=C))]).astype(np.int32)
    delta = d['gamma'] * d['total_mean'][:, None]
    x = d['counts']
    per_v = (gammaln(x.sum(axis=2) + 1.0) - gammaln(x + 1.0).sum(axis=2)).sum(axis=1)
    mult = np.array([per_v[off[c]:off[c + 1]].sum() for c in range(C)])
    lp = np.arange(2) * np.log(0.01)
    prior = lp - np.log(np.exp(lp).sum())
    rng = np.random.default_rng(0)
    eta0 = (rng.random((C, G)) < 0.5).astype(np.int32)
    tau0 = np.zeros((x.shape[0], G, 4), dtype=np.int64)
    np.put_along_axis(tau0, rng.integers(0, 4, size=(x.shape[0], G))[..., None], 1, axis=2)
    dev = _lib.Genes(0)
    dev.set_data(x, off, d['cov'])
    dev.set_model(d['gamma'], d['epsilon'], np.ascontiguousarray(delta.T), 2, prior, -gammaln(d['cov'] + 1.0).sum(axis=1), mult)
    dev.set_state(eta0, tau0)
    dev.seed(1)
    dev.update(3)                                             # warm-up
    t0 = time.perf_counter()
    dev.update(args.iters)
    dt = time.perf_counter() - t0
    out = {"metric": "accessory-gene copy-number updates/s (batched Eta_Sampler.update)", "genes": C, "samples": S,
           "haplotypes": G, "variant_rows": int(x.shape[0]), "iters": args.iters, "ms_per_iter": 1e3 * dt / args.iters,
           "value": C * G * args.iters / dt, "unit": "gene*haplotype updates/s"}
    if args.cpu_genes:
        from oracle import cbind, ref_genes as rg
        n = min(args.cpu_genes, C)
        cbind.initRNG(); cbind.setRNG(1)
        variants = [np.ascontiguousarray(x[off[c]:off[c + 1]]) for c in range(n)]
        taus = [np.ascontiguousarray(tau0[off[c]:off[c + 1]]) for c in range(n)]
        eta = eta0[:n].astype(np.int64)
        t0 = time.perf_counter()
        rg.eta_update_reference_order(np.random.RandomState(1), eta, taus, variants, d['cov'][:n], d['gamma'], d['epsilon'],
                                      np.ascontiguousarray(delta.T), prior, 2, np.zeros_like(eta), np.zeros(n))
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n * G * 2 / cdt, "unit": out["unit"], "cores": 1, "kind": "port",
                               "sample": "%d genes x 2 iterations, oracle/ref_genes.py (numpy + C tau sweep)" % n}
        out["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
