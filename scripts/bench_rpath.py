#!/usr/bin/env python3
"""Row f1 as a workflow (VERDICT r4 "missing" 4): `desman <table> -g G -i I -r R` on a synthetic V x S table -- the fit on R random positions,
then factorize_tau (gamma fixed) over the V - R positions the sampler did not see and 2 x I tau-only sweeps (updateTau) over them
(bin/desman:181-206, Init_NMFT.py:134-149,192-205, HaploSNP_Sampler.py:383-407) -- phase by phase: host + GPU wall times by cProfile
cumulative times, factorize_tau's updates run and us per update from the log, us per updateTau sweep; then the rocprofv3 averages of the
sweep and generator launches of an updateTau run of the same shape (skipped with --no-trace or without rocprofv3).
usage: bench_rpath.py [--V 50000] [--S 96] [-g 8] [-i 500] [-r 1000] [--out gpurun_out/rpath.json]"""
import argparse
import cProfile
import json
import logging
import os
import pstats
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import pandas as p  # noqa: E402

from desman_amd import cli  # noqa: E402
from desman_amd.synth import synth_counts  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--V", type=int, default=50000)
ap.add_argument("--S", type=int, default=96)
ap.add_argument("-g", "--genomes", type=int, default=8)
ap.add_argument("-i", "--iters", type=int, default=500)
ap.add_argument("-r", "--random_select", type=int, default=1000)
ap.add_argument("--strains", type=int, default=None, help="strains the table is generated from (default: -g)")
ap.add_argument("--no-trace", action="store_true")
ap.add_argument("--out", default="gpurun_out/rpath.json")
a = ap.parse_args()
V, S, G = a.V, a.S, a.genomes
counts, _, _ = synth_counts(V, S, a.strains or G, seed=1234)
cols = ["Position"] + ["S%d-%s" % (s, b) for s in range(S) for b in "ACGT"]
data = np.concatenate([np.arange(V)[:, None] * 7 + 3, counts.reshape(V, S * 4)], axis=1)
df = p.DataFrame(data, index=["contig%d" % (v // 50) for v in range(V)], columns=cols)
KEYS = {"read_table": ("cli.py", "_read_table"), "variant_filter": ("Variant_Filter.py", "__init__"), "select_random": ("Variant_Filter.py", "select_Random"),
        "fit_nmft_factorize": ("Init_NMFT.py", "factorize"), "fit_gibbs_update": ("HaploSNP_Sampler.py", "update"),
        "fit_remove_degenerate": ("HaploSNP_Sampler.py", "removeDegenerate"),
        "rest_factorize_tau": ("Init_NMFT.py", "factorize_tau"), "rest_update_tau": ("HaploSNP_Sampler.py", "updateTau"),
        "rest_assign_total": ("cli.py", "_assign_rest"), "fit_total": ("cli.py", "_fit"), "report_fit": ("cli.py", "_report"),
        "out_pred_fit": ("Output_Results.py", "outPredFit"), "out_collated_tau": ("Output_Results.py", "output_collated_Tau")}
res = dict(V=V, S=S, G=G, iters=a.iters, random_select=a.random_select, strains=a.strains or G)
with tempfile.TemporaryDirectory() as d:
    freq = os.path.join(d, "syn.freq")
    df.to_csv(freq)
    cli.main([freq, "-g", "3", "-i", "5", "-r", "200", "-o", os.path.join(d, "warm")])       # warm: library load, table cache, jump tables of the generator
    for h in logging.root.handlers[:]:
        logging.root.removeHandler(h)
    out = os.path.join(d, "out")
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    cli.main([freq, "-g", str(G), "-i", str(a.iters), "-r", str(a.random_select), "-o", out])
    pr.disable()
    res["wall_s"] = time.perf_counter() - t0
    st = pstats.Stats(pr)
    ph = {}
    for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
        base = os.path.basename(fn)
        for k, (f, n) in KEYS.items():
            if base == f and name == n:
                ph[k] = ph.get(k, 0.0) + ct
    res["phases_s"] = ph
    log = open(os.path.join(out, "log_file.txt")).read().splitlines()
    ntf = [int(ln.split("NTF Iter ")[1].split(",")[0]) for ln in log if "NTF Iter " in ln]
    # the log holds two NTF traces: the fit's (factorize) and the rest's (factorize_tau): the iteration counter restarts
    cuts = [i for i in range(1, len(ntf)) if ntf[i] < ntf[i - 1]]
    res["factorize_updates"] = ntf[cuts[0] - 1] if cuts else (ntf[-1] if ntf else None)
    res["factorize_tau_updates"] = ntf[-1] if cuts else None
    if res["factorize_tau_updates"] and "rest_factorize_tau" in ph:
        res["factorize_tau_us_per_update"] = 1e6 * ph["rest_factorize_tau"] / max(res["factorize_tau_updates"], 1)
    if "rest_update_tau" in ph:
        res["update_tau_us_per_sweep"] = 1e6 * ph["rest_update_tau"] / (2.0 * a.iters)
    res["positions_rest"] = V - a.random_select
    res["files"] = sorted(os.listdir(out))
print(json.dumps({k: v for k, v in res.items() if k != "files"}, indent=1), flush=True)
if not a.no_trace and shutil.which("rocprofv3"):
    # the sweep and generator launches of updateTau at the rest's shape, in situ
    work = tempfile.mkdtemp(prefix="rpath_", dir="/tmp")
    try:
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", work, "-o", "t", "--", sys.executable,
               os.path.join(ROOT, "scripts", "prof_update_tau2.py"), str(V - a.random_select), str(S), str(G), "100"]
        r = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, TMPDIR="/tmp"), cwd=ROOT, timeout=600)
        import csv
        import glob
        import statistics as stt
        f = glob.glob(os.path.join(work, "**", "t_kernel_trace.csv"), recursive=True)
        if r.returncode == 0 and f:
            rows = sorted(csv.DictReader(open(f[0])), key=lambda x: int(x["Start_Timestamp"]))
            dur = lambda x: (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3
            tau = [x for x in rows if x["Kernel_Name"].startswith("void tau_kernel<")]
            kern = {}
            for x in rows:
                kern.setdefault(x["Kernel_Name"].split("(")[0][:48], []).append(dur(x))
            gaps = [(int(tau[i + 1]["Start_Timestamp"]) - int(tau[i]["End_Timestamp"])) / 1e3 for i in range(len(tau) - 1)]
            res["update_tau_trace"] = dict(
                sweep_launches=len(tau), sweep_us_median=stt.median([dur(x) for x in tau]) if tau else None,
                gap_to_next_sweep_us_median=stt.median(gaps) if gaps else None,
                kernels_us_avg={k: round(sum(v) / len(v), 1) for k, v in kern.items() if k.startswith(("mt_", "void tau_kernel", "void mt_"))},
                kernels_calls={k: len(v) for k, v in kern.items() if k.startswith(("mt_", "void tau_kernel", "void mt_"))},
                stdout=r.stdout.strip().splitlines()[-2:])
            print(json.dumps(res["update_tau_trace"], indent=1))
        else:
            res["update_tau_trace"] = dict(error=(r.stderr or "")[-500:])
    finally:
        shutil.rmtree(work, ignore_errors=True)
os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
json.dump(res, open(a.out, "w"), indent=1)
