#!/usr/bin/env python3
"""Calibration of bench.py's CPU baseline: t(reference update()) / t(oracle port update()) on the same slice.

Runs ONLY in the development container (it imports /root/reference through the shims of
tests/golden/make_golden.py: np.int/np.float aliases, a stub `desman` package, and a `sampletau` module backed by
oracle/ -- the reference's own C extension needs GSL and cannot be built here, so both sides use the same C tau
sweep and the ratio measures the Python-level part of the iteration, which is 99 % of it).
Writes profiles/cpu_calibration.json, which bench.py reads on the GPU box ("x reference" = ratio * "x port").

  PYTHONDONTWRITEBYTECODE=1 python scripts/calibrate_cpu_port.py [--V 400] [--iters 3]
"""
import argparse
import json
import os
import platform
import sys
import time

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--V", type=int, default=400)
    ap.add_argument("--S", type=int, default=64)
    ap.add_argument("--G", type=int, default=8)
    ap.add_argument("--iters", type=int, default=3)
    args = ap.parse_args()
    if not os.path.isdir("/root/reference"):
        sys.exit("calibrate_cpu_port.py needs /root/reference (development container only)")
    import make_golden as mg                      # installs the import shims
    inmft, hsnp, du = mg.import_reference()
    from desman_amd.synth import synth_counts
    from oracle import cbind, ref_numpy as rn
    V, S, G = args.V, args.S, args.G
    counts, _, _ = synth_counts(V, S, G, seed=1234)

    # the reference's own class, its own update() loop (HaploSNP_Sampler.py:334-365)
    rs = np.random.RandomState(0)
    smp = hsnp.HaploSNP_Sampler(counts, G, rs, max_iter=args.iters)
    cbind.initRNG(); cbind.setRNG(0)
    t0 = time.perf_counter()
    smp.update()
    t_ref = (time.perf_counter() - t0) / args.iters

    # the oracle port of the same loop (what bench.py times on the GPU box)
    rs = np.random.RandomState(0)
    gamma0, tau0 = rn.sampler_ctor_draws(rs, V, S, G)
    eta0 = 0.96 * np.eye(4) + 0.01
    cbind.setRNG(0)
    state = (tau0, gamma0, eta0)
    t0 = time.perf_counter()
    for _ in range(args.iters):
        r = rn.gibbs_update(rs, state[0], state[1], state[2], counts, 1, cbind.sample_tau)
        state = (r["tau"], r["gamma"], r["eta"])
    t_port = (time.perf_counter() - t0) / args.iters
    cbind.freeRNG()

    out = dict(reference_s_per_iter=t_ref, port_s_per_iter=t_port, reference_over_port=t_ref / t_port,
               V=V, S=S, G=G, iters=args.iters, host=platform.processor() or platform.machine(),
               cpu=_cpu_model(), threads=1,
               note="imported reference HaploSNP_Sampler.update() vs oracle/ref_numpy.gibbs_update on the same "
                    "synthetic slice, same C tau sweep (oracle) on both sides, 1 thread, development container")
    path = os.path.join(ROOT, "profiles", "cpu_calibration.json")
    json.dump(out, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
