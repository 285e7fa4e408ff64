#!/usr/bin/env python3
"""BASELINE config 5 on ONE GPU: the G-sweep g = 2..12 x 5 seeds (55 chains; scripts/runDesman.sh:15-21,
scripts/desmanflow.nf:119-136) over a synthetic V = 50 000 x S = 96 table, -i 500, through the `desman-sweep` driver
(desman_amd.chains.main: every chain is a whole `desman` run -- CSV in, NMF start, 2 x 500 Gibbs iterations, result files
out), once chain by chain and once with the five replicates of a G value batched (-b 5); then posterior-deviance model
selection (desman_amd.resolvenhap = scripts/resolvenhap.py:118-217) against the generating G.

Also checks the scheduler's cost model: per-chain wall times against chains.chain_cost, and the 8-GPU LPT plan the
driver would make -- predicted max/mean bin load vs the load the measured times give for the same assignment.

usage: bench_config5.py [--V 50000] [--S 96] [--gmin 2] [--gmax 12] [--reps 5] [-i 500] [--G-true 6] [--out gpurun_out/r03_config5.json]
"""
import argparse
import contextlib
import io
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import pandas as p  # noqa: E402

from desman_amd import chains, resolvenhap  # noqa: E402
from desman_amd.synth import synth_counts  # noqa: E402


def run_sweep(freq, a, stub, extra):
    buf = io.StringIO()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(buf):
        chains.main([freq, "--gmin", str(a.gmin), "--gmax", str(a.gmax), "--reps", str(a.reps), "-i", str(a.iters), "-o", stub] + extra)
    wall = time.perf_counter() - t0
    recs = json.loads([ln for ln in buf.getvalue().splitlines() if ln.startswith("[{")][0])
    return wall, recs


def plan_check(recs, V, S, n_bins=8, batch=1):
    """LPT plan of the driver for n_bins GPUs from chain_cost; bin loads predicted (cost units) and measured (seconds)"""
    specs = chains.sweep_specs(sorted({int(r["G"]) for r in recs}), 1 + max(int(r["seed"]) for r in recs), V, S, n_iter=500)
    wall = {(int(r["G"]), int(r["seed"])): r["wall_s"] for r in recs}
    units = chains.group_units(specs, batch)
    bins = chains.lpt_assign([sum(specs[i]["cost"] for i in u) for u in units], n_bins)
    pred = [sum(specs[i]["cost"] for ui in b for i in units[ui]) for b in bins]
    meas = [sum(wall[(specs[i]["G"], specs[i]["seed"])] for ui in b for i in units[ui]) for b in bins]
    # the best the same greedy rule could do had it known the measured times
    ub = chains.lpt_assign([sum(wall[(specs[i]["G"], specs[i]["seed"])] for i in u) for u in units], n_bins)
    best = [sum(wall[(specs[i]["G"], specs[i]["seed"])] for ui in b for i in units[ui]) for b in ub]
    # the work queue (desman-sweep --schedule queue, the default): every rank takes the next unit of the list (decreasing estimate) when
    # it is free -- simulated with the measured times
    order = sorted(range(len(units)), key=lambda u: (-sum(specs[i]["cost"] for i in units[u]), u))
    free = [0.0] * n_bins
    for u in order:
        k = min(range(n_bins), key=lambda r: (free[r], r))
        free[k] += sum(wall[(specs[i]["G"], specs[i]["seed"])] for i in units[u])
    return dict(bins=n_bins, predicted_max_over_mean=max(pred) / (sum(pred) / n_bins), measured_max_over_mean=max(meas) / (sum(meas) / n_bins),
                measured_makespan_s=max(meas), lpt_on_measured_times_makespan_s=max(best), sum_s=sum(meas),
                work_queue_makespan_s=max(free), work_queue_max_over_mean=max(free) / (sum(meas) / n_bins))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--V", type=int, default=50000)
    ap.add_argument("--S", type=int, default=96)
    ap.add_argument("--gmin", type=int, default=2)
    ap.add_argument("--gmax", type=int, default=12)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("-i", "--iters", type=int, default=500)
    ap.add_argument("--G-true", type=int, default=6)
    ap.add_argument("--out", default="gpurun_out/r03_config5.json")
    ap.add_argument("--modes", default="one,batch")
    a = ap.parse_args()
    counts, _, gamma_true = synth_counts(a.V, a.S, a.G_true, seed=1234)
    cols = ["Position"] + ["S%d-%s" % (s, b) for s in range(a.S) for b in "ACGT"]
    data = np.concatenate([np.arange(a.V)[:, None] * 7 + 3, counts.reshape(a.V, a.S * 4)], axis=1)
    df = p.DataFrame(data, index=["contig%d" % (v // 50) for v in range(a.V)], columns=cols)
    out = dict(V=a.V, S=a.S, g=[a.gmin, a.gmax], reps=a.reps, iters=a.iters, G_true=a.G_true,
               gamma_true_mean=np.sort(gamma_true.mean(axis=0))[::-1].round(4).tolist(),
               chains=(a.gmax - a.gmin + 1) * a.reps)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with tempfile.TemporaryDirectory() as d:
        freq = os.path.join(d, "syn.freq")
        t0 = time.perf_counter()
        df.to_csv(freq)
        out["csv_mb"] = os.path.getsize(freq) / 1e6
        out["csv_write_s"] = time.perf_counter() - t0
        for mode in a.modes.split(","):
            stub = os.path.join(d, mode)
            extra = {"one": ["-c", "1"], "batch": ["-b", str(a.reps), "-c", "1"], "batch2": ["-b", str(a.reps), "-c", "2"],
                     "threads2": ["-c", "2"], "threads4": ["-c", "4"]}[mode]
            wall, recs = run_sweep(freq, a, stub, extra)
            per_g = {}
            for r in recs:
                per_g.setdefault(int(r["G"]), []).append(r["wall_s"])
            cost = {g: chains.chain_cost(a.V, a.S, g, n_iter=a.iters) for g in per_g}
            tot_w, tot_c = sum(np.mean(v) for v in per_g.values()), sum(cost.values())
            res = dict(wall_s=wall, failed=int(sum(r["failed"] for r in recs)),
                       per_G_mean_chain_wall_s={g: float(np.mean(v)) for g, v in sorted(per_g.items())},
                       per_G_share_measured={g: float(np.mean(v) / tot_w) for g, v in sorted(per_g.items())},
                       per_G_share_cost_model={g: cost[g] / tot_c for g in sorted(per_g)},
                       lpt_8_gpus=plan_check(recs, a.V, a.S, 8, a.reps if mode.startswith("batch") else 1))
            dev = open(stub + "_Dev.csv").read()
            res["dev_csv"] = dev
            rows = [ln.split(",") for ln in dev.strip().split("\n")[1:]]
            by = {}
            for h, g, lp, dv in rows:
                if h == g:
                    by.setdefault(int(h), []).append(float(dv))
            res["mean_dev_per_G"] = {g: float(np.mean(v)) for g, v in sorted(by.items())}
            pick = resolvenhap.resolve(stub, write=False)
            res["resolvenhap"] = None if pick is None else dict(bestG=int(pick[0]), n_strains=int(pick[1]), replicate=int(pick[2]),
                                                                 mean_err=float(pick[3]))
            out[mode] = res
            shutil.copy(stub + "_Dev.csv", os.path.splitext(a.out)[0] + "_%s_Dev.csv" % mode)
            print(mode, "wall %.1f s" % wall, "pick", res["resolvenhap"], flush=True)
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk != "dev_csv"}) for k, v in out.items()}))


if __name__ == "__main__":
    main()
