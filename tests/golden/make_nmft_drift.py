#!/usr/bin/env python3
"""How far do independent evaluations of Init_NMFT.factorize() drift apart over the 5000 updates the reference really
runs on COG0015 (Init_NMFT.py:106: the 1e-5 stop never fires there)?

Three evaluations of the same iteration from the same start (RandomState(23724839), the default seed of bin/desman):
  ref    the imported reference (numpy / BLAS dot products, Python renormalisation loop)      -- development container only
  orc    oracle/desman_oracle.c (plain C loops, another summation order)
  dev    the HIP kernels (MFMA contractions, fixed-order reductions)                           -- GPU box only
plus `pert`: orc started from the same factors with every tau entry moved by one unit in the last place -- what the
iteration itself does to a rounding-sized difference, whoever computes it.

  --reference   (development container) run ref, orc, pert; write tests/golden/drift_nmft_cog0015.npz: the reference's
                factors at the checkpoints (the fixture test_gpu_host.py compares the device with) and the pairwise table
  --device      (GPU box) run dev at the same checkpoints -> gpurun_out/nmft_drift_device.npz
  --compare F   (development container) print the pairwise table incl. the device run recorded in F

No reference source is copied; the fixture holds inputs' seed and output arrays only.
"""
import argparse
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

CHECKPOINTS = (100, 1000, 5000)
SEED, G = 23724839, 5


def load_counts():
    d = np.load(os.path.join(HERE, "cog0015_counts.npz"))
    return np.ascontiguousarray(d["counts"].astype(np.int64))


def diffs(a, b, floor=1e-3):
    """(max |a-b|, max relative difference over the entries above `floor`)"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    big = np.maximum(np.abs(a), np.abs(b)) > floor
    rel = np.abs(a - b)[big] / np.maximum(np.abs(a), np.abs(b))[big]
    return float(np.abs(a - b).max()), float(rel.max()) if rel.size else 0.0


def argmax_flips(ta, tb, V):
    """fraction of (variant, haplotype) columns whose arg-max base differs (what get_tau would report)"""
    a = ta.reshape(4, V, -1).argmax(axis=0)
    b = tb.reshape(4, V, -1).argmax(axis=0)
    return float((a != b).mean())


def run_oracle(F, tau0, gam0):
    from oracle import cbind
    out, tau, gam, done = {}, tau0.copy(), gam0.copy(), 0
    # factorize() = _adjustment, then (div_update, _adjustment) per update: resuming from a checkpoint re-applies an
    # idempotent clamp, so running k1, then k2 - k1 more updates IS the k2-update run
    for k in CHECKPOINTS:
        n, _ = cbind.nmft_factorize(F, tau, gam, max_iter=k - done, min_change=0.0)
        assert n == k - done
        done = k
        out[k] = (tau.copy(), gam.copy())
    return out


def run_reference(counts, rs):
    sys.path.insert(0, HERE)
    import make_golden as mg
    inmft, _, _ = mg.import_reference()
    import math
    nm = inmft.Init_NMFT(counts, G, rs)
    nm.random_initialize()
    tau0, gam0 = nm.tau.copy(), nm.gamma.copy()
    nm._adjustment()
    out, divl, div, it = {}, 0.0, nm.div_objective(), 0
    while it < nm.max_iter and math.fabs(divl - div) > nm.min_change:        # Init_NMFT.py:106
        nm.div_update()
        nm._adjustment()
        divl, div = div, nm.div_objective()
        it += 1
        if it in CHECKPOINTS:
            out[it] = (nm.tau.copy(), nm.gamma.copy())
            print("reference: %d updates, div = %.6f" % (it, div), flush=True)
    assert it == 5000, "the stop test fired after %d updates" % it
    return tau0, gam0, out, np.asarray(nm.freq_matrix)


def table(runs, V):
    names = sorted(runs)
    rows = []
    for i, a in enumerate(names):
        for b in names[i + 1:]:
            for k in CHECKPOINTS:
                (ta, ga), (tb, gb) = runs[a][k], runs[b][k]
                dt, dg = diffs(ta, tb), diffs(ga, gb)
                rows.append((a, b, k, dt[0], dt[1], dg[0], dg[1], argmax_flips(ta, tb, V)))
    return rows


def print_table(rows):
    print("%-5s %-5s %6s  %11s %11s  %11s %11s  %9s" % ("a", "b", "upd", "tau max|d|", "tau rel", "gam max|d|", "gam rel", "argmax"))
    for r in rows:
        print("%-5s %-5s %6d  %11.3e %11.3e  %11.3e %11.3e  %9.2e" % r)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", action="store_true")
    ap.add_argument("--device", action="store_true")
    ap.add_argument("--compare")
    a = ap.parse_args()
    counts = load_counts()
    V = counts.shape[0]
    if a.device:
        from desman_amd.Init_NMFT import Init_NMFT
        nm = Init_NMFT(counts, G, np.random.RandomState(SEED))
        nm.random_initialize()
        tau0, gam0 = nm.tau.copy(), nm.gamma.copy()
        rec, done = {"tau0": tau0, "gam0": gam0}, 0
        nm._push()
        for k in CHECKPOINTS:
            n, tr = nm._ctx.nmft_factorize(k - done, 0.0, fix_gamma=False)
            assert n == k - done
            done = k
            t, g = nm._ctx.nmft_get()
            rec["tau_%d" % k], rec["gam_%d" % k], rec["div_%d" % k] = t, g, tr[-1]
            print("device: %d updates, div = %.6f" % (k, tr[-1]), flush=True)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", "nmft_drift_device.npz"), **rec)
        return
    fix = os.path.join(HERE, "drift_nmft_cog0015.npz")
    if a.reference:
        tau0, gam0, ref, F = run_reference(counts, np.random.RandomState(SEED))
        F = np.ascontiguousarray(F)
        orc = run_oracle(F, tau0, gam0)
        tp = np.nextafter(tau0, np.inf)                                   # every entry one unit in the last place up
        pert = run_oracle(F, tp, gam0)
        rows = table({"ref": ref, "orc": orc, "pert": pert}, V)
        print_table(rows)
        rec = {"seed": SEED, "G": G, "tau0": tau0, "gam0": gam0,
               "table": np.array([(r[0] + ":" + r[1], r[2]) + r[3:] for r in rows], dtype=object).astype(str)}
        for k in CHECKPOINTS:
            rec["ref_tau_%d" % k], rec["ref_gam_%d" % k] = ref[k]
        np.savez_compressed(fix, **rec)
        return
    if a.compare:
        from oracle import cbind
        z, d = np.load(fix), np.load(a.compare)
        assert np.array_equal(z["tau0"], d["tau0"]) and np.array_equal(z["gam0"], d["gam0"]), "different start"
        F = cbind.nmft_freq(counts)
        runs = {"ref": {k: (z["ref_tau_%d" % k], z["ref_gam_%d" % k]) for k in CHECKPOINTS},
                "dev": {k: (d["tau_%d" % k], d["gam_%d" % k]) for k in CHECKPOINTS},
                "orc": run_oracle(F, z["tau0"], z["gam0"]),
                "pert": run_oracle(F, np.nextafter(z["tau0"], np.inf), z["gam0"])}
        print_table(table(runs, V))


if __name__ == "__main__":
    main()
