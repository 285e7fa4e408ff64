#!/usr/bin/env python3
"""Headline benchmark: Gibbs iterations/s on synthetic V=10k x S=64, G=8 (BASELINE.json
configs[2]); one independent chain per GPU (weak scaling), RCCL gather of the fit records.

  python bench.py --gpus N --steps 500 --warmup 50        (N > 1: starts its own N ranks, desman_amd/launch.py)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the driver's form)

In both forms the world MUST be N ranks, one per GPU: anything else exits with status 2 and a message, no JSON line.

Prints ONE JSON line on rank 0.  A "step" is one full Gibbs iteration (auxiliary-count
pass, gamma/eta draws, tau sweep, log-posterior, MAP/trace bookkeeping) of one chain.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def _cpu_chain(args):
    """One CPU chain of the baseline: `its` full Gibbs iterations of the oracle port on the V=Vs slice."""
    Vs, S, G, its, seed = args
    from desman_amd.synth import synth_counts
    from oracle import cbind, ref_numpy as rn
    counts, _, _ = synth_counts(Vs, S, G, seed=1234)
    rs = np.random.RandomState(seed)
    gamma0, tau0 = rn.sampler_ctor_draws(rs, Vs, S, G)
    state = (tau0, gamma0, 0.96 * np.eye(4) + 0.01)
    cbind.initRNG(); cbind.setRNG(seed)
    t0 = time.perf_counter()
    for _ in range(its):
        r = rn.gibbs_update(rs, state[0], state[1], state[2], counts, 1, cbind.sample_tau)
        state = (r["tau"], r["gamma"], r["eta"])
    dt = time.perf_counter() - t0
    cbind.freeRNG()
    return dt


def cpu_baseline_main(S, G):
    """`python bench.py --cpu-baseline-only`: the CPU path (oracle = restatement of the reference: Python-level
    sampleMu / likelihood loops + C tau sweep) timed on a bounded sample of the same workload -- 1 thread (the
    reference is single-threaded by construction) and all cores (one independent chain per core, the only
    parallelism the reference has: scripts/runDesman.sh:15-21).  Runs in its own process: no HIP runtime here."""
    import multiprocessing as mp
    Vs = 400
    its1 = 5
    dt1 = _cpu_chain((Vs, S, G, its1, 0))
    cores = sorted(os.sched_getaffinity(0))
    n = len(cores)
    its_all = 1
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(n) as pool:
        dts = pool.map(_cpu_chain, [(Vs, S, G, its_all, k) for k in range(n)])
    wall = time.perf_counter() - t0
    calib = {}
    try:
        calib = json.load(open(os.path.join(ROOT, "profiles", "cpu_calibration.json")))
    except OSError:
        pass
    out = dict(value=Vs * S * its1 / dt1, unit="V*S updates/s", cores=1, kind="port",
               sample="%d full Gibbs iterations of oracle/ref_numpy.gibbs_update (+ C tau sweep) on a V=%d slice of "
                      "the same S=%d, G=%d workload, 1 thread, %.1f s" % (its1, Vs, S, G, dt1),
               cpu=_cpu_model(),
               all_cores=dict(value=n * Vs * S * its_all / wall, unit="V*S updates/s", cores=n,
                              sample="%d independent chains (one per core in sched_getaffinity), %d iterations each of the "
                                     "same slice, wall %.1f s (slowest chain %.1f s)" % (n, its_all, wall, max(dts))),
               # t(reference update()) / t(this port) on the same slice: written by scripts/calibrate_cpu_port.py in the
               # development container (the only place /root/reference exists)
               calibration_reference_over_port=calib.get("reference_over_port"),
               calibration_source="profiles/cpu_calibration.json (scripts/calibrate_cpu_port.py: %s)" % calib.get("cpu", "?"))
    print(json.dumps(out))


def cpu_baseline(S, G):
    """Runs the CPU legs in a child process (fork-based multiprocessing next to a live HIP runtime is unsafe)."""
    import subprocess
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--S", str(S), "--G", str(G)],
                       capture_output=True, text=True, timeout=600, env=env)
    if r.returncode != 0:
        raise RuntimeError("cpu baseline failed: " + r.stderr[-2000:])
    return json.loads(r.stdout.strip().splitlines()[-1])


TRAFFIC_DB = os.path.join(ROOT, "profiles", "pmc_traffic_by_shape.json")
PROF_NMFT_UPDATES = 200          # NMFT updates scripts/prof_gibbs.py runs before its Gibbs iterations (one factorize call)


def nmft_pmc_summary(tj, us_per_update):
    """NMFT figures out of a pmc_passes() record: HBM-side bytes, matrix-core and VALU instructions PER UPDATE, summed over
    every nmft_* kernel of the profiled run (the persistent kernel is ONE launch for all its updates, the three-launch path
    three launches per update: dispatches x per-launch mean / updates covers both), and the share of the launch's SIMD cycles
    the MFMA pipe was busy: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x update time x 2.4 GHz)."""
    tot = dict(bytes=0.0, mfma=0.0, busy=0.0, valu=0.0)
    names = []
    for k, r in (tj or {}).items():
        kk = k[5:] if k.startswith("void ") else k
        if not kk.startswith("nmft_"):
            continue
        n = r.get("dispatches") or 0
        names.append(kk)
        tot["bytes"] += n * (r.get("bytes_per_launch") or 0.0)
        tot["mfma"] += n * (r.get("mfma_insts") or 0.0)
        tot["busy"] += n * (r.get("mfma_busy_cycles") or 0.0)
        tot["valu"] += n * (r.get("valu_insts") or 0.0)
    if not names or tot["bytes"] <= 0:
        return None
    per = {k: v / PROF_NMFT_UPDATES for k, v in tot.items()}
    cyc = 1024 * us_per_update * 1e-6 * 2.4e9
    return dict(traffic=per["bytes"], mfma_insts_per_update=per["mfma"], valu_insts_per_update=per["valu"],
                mfma_busy_frac=per["busy"] / cyc if cyc > 0 else None, kernels=sorted(names))


def shape_key(V, S, G, depth):
    return "%d,%d,%d,%g" % (V, S, G, depth)


def pmc_passes(V, S, G, depth, iters=30, save=False):
    """`bench.py --pmc`: HBM-side traffic and VALU instruction counts per launch of this workload's kernels, measured now: three
    separate `rocprofv3 --pmc` passes (FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU -- one counter group per pass, kernel trace only, as
    MI355X_MICROARCH.md prescribes; FETCH_SIZE x 2 is its gfx950 correction for wide coalesced reads) over scripts/prof_gibbs.py,
    the benchmark's chain at this shape.  Returns {kernel name: {bytes_per_launch, read_bytes_corrected, write_bytes, valu_insts}}
    and, with save (`--pmc-save`), stores it in the tracked profiles/pmc_traffic_by_shape.json under the shape key, where runs with
    --no-pmc find it; the scratch copy under gpurun_out/ is written in any case (a default run leaves `git status` clean)."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    env = dict(os.environ, TMPDIR="/tmp")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    work = tempfile.mkdtemp(prefix="dsm_pmc_", dir="/tmp")
    try:
        for ctr in (["FETCH_SIZE"], ["WRITE_SIZE"],
                    ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES"]):
            d = os.path.join(work, ctr[0])
            cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + ctr + ["--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                   os.path.join(ROOT, "scripts", "prof_gibbs.py"), str(iters), str(V), str(S), str(G), str(depth)]
            r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd="/tmp", timeout=900)
            if r.returncode != 0:
                raise RuntimeError("rocprofv3 pass %s failed: %s" % (ctr, r.stderr[-1500:]))
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    agg[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    finally:
        shutil.rmtree(work, ignore_errors=True)
    out = {}
    for k, dd in agg.items():
        mean = {c: sum(v) / len(v) for c, v in dd.items()}
        fe, wr = mean.get("FETCH_SIZE", 0.0), mean.get("WRITE_SIZE", 0.0)            # KiB per launch
        out[k] = dict(fetch_kib_raw=fe, write_kib_raw=wr, read_bytes_corrected=2.0 * fe * 1024.0, write_bytes=wr * 1024.0,
                      bytes_per_launch=2.0 * fe * 1024.0 + wr * 1024.0, valu_insts=mean.get("SQ_INSTS_VALU"),
                      valu_active_cycles=mean.get("SQ_ACTIVE_INST_VALU"), sq_busy_cycles=mean.get("SQ_BUSY_CYCLES"),
                      dispatches=max(len(v) for v in dd.values()),
                      mfma_insts=mean.get("SQ_INSTS_MFMA"), mfma_busy_cycles=mean.get("SQ_VALU_MFMA_BUSY_CYCLES"))
    db = {}
    if os.path.exists(TRAFFIC_DB):
        db = json.load(open(TRAFFIC_DB))
    db[shape_key(V, S, G, depth)] = out
    for path in ((TRAFFIC_DB,) if save else ()) + (os.path.join(ROOT, "gpurun_out", "pmc_traffic_by_shape.json"),):
        try:
            os.makedirs(os.path.dirname(path), exist_ok=True)
            json.dump(db, open(path, "w"), indent=1, sort_keys=True)
        except OSError:
            pass
    return out


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip() + " x%d" % os.cpu_count()
    except OSError:
        pass
    return "unknown"


def bench_genes(n_genes, S, G, vmax, iters, cpu_genes):
    """Row f4 (accessory-gene sampler), separate from the headline metric: `python bench.py --workload genes`.
    Batched (rng='philox') iterations of Eta_Sampler.update over n_genes genes on one GPU; the CPU leg is the oracle's
    reference-order loop (numpy + C tau sweep, 1 thread) on a bounded sample of the genes."""
    from scipy.special import gammaln
    from desman_amd import _lib
    from desman_amd.synth import synth_genes
    C = n_genes
    d = synth_genes(C, S, G, seed=5, vmax=vmax, mean_lo=2.0, mean_hi=20.0)
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
    dev.update(iters)
    dt = time.perf_counter() - t0
    out = {"metric": "accessory-gene copy-number updates/s (batched Eta_Sampler.update)", "genes": C, "samples": S,
           "haplotypes": G, "variant_rows": int(x.shape[0]), "iters": iters, "ms_per_iter": 1e3 * dt / iters,
           "value": C * G * iters / dt, "unit": "gene*haplotype updates/s", "n_gpus": 1, "data": "synthetic"}
    if cpu_genes:
        from oracle import cbind, ref_genes as rg
        n = min(cpu_genes, C)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--V", type=int, default=10000)
    ap.add_argument("--S", type=int, default=64)
    ap.add_argument("--G", type=int, default=8)
    ap.add_argument("--rng", choices=["mt19937", "philox"], default="mt19937")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="internal: print the cpu_baseline object and exit")
    ap.add_argument("--true-G", type=int, default=None, help="strains the synthetic table is generated from (default: --G).  A G-sweep fits "
                    "most of its chains with too few or too many haplotypes (BASELINE config 5: g = 2..12 on one table): their iterations cost more")
    ap.add_argument("--stats-spec", type=int, default=0, choices=[0, 1, 2, 3, 4], help="force a mu/E specification (0 = the shape rule; "
                    "4 = stage 1 over tau words)")
    ap.add_argument("--depth-scale", type=float, default=1.0, help="multiply the mean read depths of the synthetic tensor")
    ap.add_argument("--chains-per-gpu", type=int, default=1,
                    help="also time K concurrent chains on the GPU (extra key; the headline stays one chain per GPU)")
    ap.add_argument("--batch", type=int, default=None, help="extra key `batch`: K chains of this shape in one set of launches "
                    "(dsm_batch_gibbs_update); default 4 chains x at most 100 steps on a single-GPU run, 0/1 = off")
    ap.add_argument("--no-nmft", action="store_true")
    ap.add_argument("--repeats", type=int, default=7, help="the timed call of exactly --steps iterations is made this many times, each "
                    "between its own barrier + synchronize; ms_per_step is the median call (ms_per_step_repeats has them all)")
    ap.add_argument("--pmc", dest="pmc", action="store_true", default=None, help="measure roofline.traffic now: three extra rocprofv3 "
                    "--pmc passes of this workload (about a minute).  Default: on for a single-GPU run when rocprofv3 is on the PATH")
    ap.add_argument("--pmc-save", action="store_true", help="also record the measured traffic in the tracked profiles/pmc_traffic_by_shape.json "
                    "(what --no-pmc runs read); without it only gpurun_out/ is written")
    ap.add_argument("--no-pmc", dest="pmc", action="store_false", help="skip the PMC passes: roofline.traffic is then read from the "
                    "record `bench.py --pmc` left for this shape in profiles/pmc_traffic_by_shape.json (said so in traffic_source)")
    ap.add_argument("--counts-npz", default=None,
                    help="real data instead of the synthetic tensor: an .npz with `counts` [V,S,4] (tests/golden/cog0015_counts.npz "
                         "= the reference's complete_example table after its sample filter); V and S come from the file")
    ap.add_argument("--lrt-filter", action="store_true",
                    help="with --counts-npz: run the reference's `-f` likelihood-ratio variant filter first (lrt_kernel)")
    ap.add_argument("--workload", choices=["gibbs", "genes"], default="gibbs",
                    help="gibbs = the headline metric (default); genes = the accessory-gene sampler (row f4)")
    ap.add_argument("--genes", type=int, default=2000)
    ap.add_argument("--vmax", type=int, default=20)
    ap.add_argument("--cpu-genes", type=int, default=40)
    args = ap.parse_args()
    if args.cpu_baseline_only:
        return cpu_baseline_main(args.S, args.G)
    if args.workload == "genes":
        return bench_genes(args.genes, 32 if args.S == 64 else args.S, 6 if args.G == 8 else args.G, args.vmax,
                           50 if args.steps == 500 else args.steps, 0 if args.no_cpu_baseline else args.cpu_genes)

    # the world is exactly --gpus ranks (one per GPU) or the run stops here: started by torch.distributed.run with another
    # WORLD_SIZE -> exit 2; plain process with --gpus N > 1 -> this process becomes that launch (desman_amd/launch.py;
    # the fan-out of scripts/runDesman.sh:15-21); fewer than N devices -> exit 2.  Never a smaller run under a bigger label.
    from desman_amd import launch
    rank, local_rank, world, under_launcher = launch.ensure_world(args.gpus, sys.argv[1:], script=__file__)
    import torch
    dist = None
    backend = launch.dist_backend()                              # "nccl" (RCCL) unless DESMAN_DIST_BACKEND=gloo (the two-ranks-on-one-GPU rehearsal)
    dev = launch.bind_device(local_rank, torch.cuda.device_count(), "bench.py") if under_launcher else local_rank
    if under_launcher:                                           # any N >= 1: communicator, barrier, MAX all-reduce, gather
        import torch.distributed as dist
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo")
        if dist.get_world_size() != args.gpus:                   # belt and braces: the process group agrees with --gpus
            launch._die("bench.py", "process group of %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    torch.cuda.set_device(dev)

    from desman_amd import _lib
    from desman_amd.synth import synth_counts
    V, S, G = args.V, args.S, args.G
    data_label = "synthetic"
    if args.counts_npz:
        counts = np.ascontiguousarray(np.load(args.counts_npz)["counts"].astype(np.int64))
        data_label = "real: %s (%d positions)" % (os.path.basename(args.counts_npz), counts.shape[0])
        if args.lrt_filter:
            import pandas as pd
            from numpy.random import RandomState
            from desman_amd.Variant_Filter import Variant_Filter
            tab = pd.DataFrame(np.concatenate([np.arange(counts.shape[0])[:, None], counts.reshape(counts.shape[0], -1)], axis=1))
            flt = Variant_Filter(tab, randomState=RandomState(0), optimise=True, threshold=3.84, min_coverage=5.0, qvalue_cutoff=1.0e-3)
            flt.device = dev
            counts = np.ascontiguousarray(flt.get_filtered_VariantsLogRatio().astype(np.int64))
            data_label += ", -f filter kept %d" % counts.shape[0]
        V, S = counts.shape[0], counts.shape[1]
    else:
        counts, _, _ = synth_counts(V, S, args.true_G or G, seed=1234 + rank, depth_scale=args.depth_scale)   # one independent chain per GPU
        if args.true_G and args.true_G != G:
            data_label = "synthetic, generated from %d strains" % args.true_G
    ctx = _lib.Context(dev)
    ctx.set_counts(counts)
    ctx.seed(rank)                                               # sampler seeds 0..N-1 (scripts/runDesman.sh:15-19)
    ctx.set_tau_rng(_lib.RNG_MT19937 if args.rng == "mt19937" else _lib.RNG_PHILOX)
    if args.stats_spec:
        ctx.force_stats_spec(args.stats_spec)

    # NMFT initialisation (untimed here; reported separately)
    rs = np.random.RandomState(rank)
    nmft = None
    alpha0 = 0.01
    if G > 1:
        gam0 = np.ascontiguousarray(rs.dirichlet(np.full(G, alpha0), size=S).T)
    else:
        gam0 = np.ones((G, S))
    d = rs.dirichlet(np.full(4, alpha0), size=V * G).reshape(V, G, 4)
    tau0 = np.ascontiguousarray(np.transpose(d, (2, 0, 1)).reshape(4 * V, G))
    ctx.nmft_set(tau0, gam0)
    if not args.no_nmft:
        n_nm = 200
        ctx.nmft_factorize(max_iter=5, min_change=0.0)           # warm
        ctx.nmft_set(tau0, gam0)
        t0 = time.perf_counter()
        n_done, tr = ctx.nmft_factorize(max_iter=n_nm, min_change=0.0)
        t_nm = time.perf_counter() - t0
        nmft = dict(iters=n_done, ms_per_iter=1e3 * t_nm / max(n_done, 1), div_last=float(tr[-1]))
        ctx.set_timing(True)
        ctx.nmft_factorize(max_iter=50, min_change=0.0)
        tmn = ctx.get_timing()
        ctx.set_timing(False)
        nmft["kernels_us"] = {k: 1e3 * ms / max(n, 1) for k, (ms, n) in tmn.items() if n and k.startswith("nmft")}
        # algorithmic bytes per NMFT update: two passes over f64 F + four over tau
        nmft["algorithmic_bytes_per_iter"] = 2 * 4 * V * S * 8 + 4 * 4 * V * G * 8
        nmft["achieved_GBps"] = nmft["algorithmic_bytes_per_iter"] / (nmft["ms_per_iter"] * 1e-3) / 1e9
        # the one bandwidth-shaped kernel of the path; at this size F + tau (23 MB) live in the
        # 256 MB Infinity Cache, so the byte stream is MALL/L2 traffic, not HBM
        nmft["roofline"] = dict(bound="hbm", achieved=nmft["achieved_GBps"], peak=8000.0, unit="GB/s",
                                frac=nmft["achieved_GBps"] / 8000.0)
        # factorize_tau (gamma fixed: Init_NMFT.py:134-149, what the `-r` workflow runs over every position the sampler did not see):
        # one fused pass per update -- a pass over F, tau read and written
        _, g_fit = ctx.nmft_get()
        ctx.nmft_set(tau0, g_fit)
        ctx.nmft_factorize(max_iter=5, min_change=0.0, fix_gamma=True)      # warm
        ctx.nmft_set(tau0, g_fit)
        t0 = time.perf_counter()
        n_ft, tr_ft = ctx.nmft_factorize(max_iter=n_nm, min_change=0.0, fix_gamma=True)
        t_ft = time.perf_counter() - t0
        ft_bytes = 4 * V * S * 8 + 2 * 4 * V * G * 8
        nmft["factorize_tau"] = dict(iters=n_ft, ms_per_iter=1e3 * t_ft / max(n_ft, 1), div_last=float(tr_ft[-1]),
                                     algorithmic_bytes_per_iter=ft_bytes,
                                     achieved_GBps=ft_bytes / (t_ft / max(n_ft, 1)) / 1e9)
        ctx.nmft_set(tau0, gam0)
        ctx.nmft_factorize(max_iter=n_nm, min_change=0.0)
    tau_init = ctx.nmft_get_tau()
    _, gam = ctx.nmft_get()
    eta0 = 0.96 * np.eye(4) + 0.01
    ctx.set_state(tau_init, np.ascontiguousarray(gam.T), eta0)

    def fence():
        ctx.sync()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()                                          # first collective of the job: communicator set-up happens here, not
    ctx.gibbs_update(args.warmup)                    # between the warm-up steps and the timed ones (an idle GPU clocks down)
    # the timed call: EXACTLY --steps iterations between barrier + synchronize on both sides, the job's time = MAX over ranks.
    # It is made --repeats times (the process-to-process spread of one 2 ms call is +-4 %): ms_per_step is the MEDIAN call.
    rep_dt, rep_mine = [], []
    for _ in range(max(args.repeats, 1)):
        fence()
        t0 = time.perf_counter()
        ctx.gibbs_update(args.steps)                 # returns after the library's stream has drained
        torch.cuda.synchronize()
        dt_r = time.perf_counter() - t0              # this rank: common start (barrier) -> its own K steps done
        fence()                                      # closing barrier + synchronize; the job's time is the MAX over ranks (below),
        rep_mine.append(dt_r)                        # i.e. what the closing barrier waits for, without the collective's own latency
        if dist is not None:
            t = torch.tensor([dt_r], device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt_r = float(t.item())
        rep_dt.append(dt_r)
    dt = float(np.median(rep_dt))
    # which device every rank bound and how long ITS steps took (median call): the per-rank view behind the MAX
    me = launch.bound_device_record(rank, dev)
    me["local_rank"] = int(local_rank)
    me["ms_per_step"] = 1e3 * float(np.median(rep_mine)) / args.steps
    ranks = [me]
    if dist is not None:
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    tr = ctx.get_trace()
    star = ctx.get_star()
    # the one collective of the path: gather every chain's fit record (lp_star, mean deviance)
    rec = [float(G), float(G), float(rank), star["lp"], -2.0 * float(tr["ll"].mean()), float(args.steps)]
    fits = [rec]
    if dist is not None:
        mine = torch.tensor(rec, device="cuda" if backend == "nccl" else "cpu", dtype=torch.float64)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        fits = [x.tolist() for x in allr]

    # several chains on one GPU at once (own context / streams / host thread each): how much of the per-launch
    # latency (Dirichlet / stage 2 / finalize / launch gaps) concurrent chains recover (scripts/runDesman.sh:15-21)
    multi = None
    if args.chains_per_gpu > 1:
        from concurrent.futures import ThreadPoolExecutor
        K = args.chains_per_gpu
        ctxs = []
        for k in range(K):
            c2 = _lib.Context(dev)
            c2.set_counts(counts)
            c2.seed(1000 + k)
            c2.set_tau_rng(_lib.RNG_MT19937 if args.rng == "mt19937" else _lib.RNG_PHILOX)
            c2.set_state(tau_init, np.ascontiguousarray(gam.T), eta0)
            c2.gibbs_update(max(args.warmup, 5))
            ctxs.append(c2)
        with ThreadPoolExecutor(K) as ex:
            t0 = time.perf_counter()
            list(ex.map(lambda c2: c2.gibbs_update(args.steps), ctxs))
            dtk = time.perf_counter() - t0
        for c2 in ctxs:
            c2.close()
        multi = dict(chains=K, ms_per_step_per_chain=1e3 * dtk / args.steps, value=K * V * S * args.steps / dtk,
                     unit="V*S updates/s", speedup_vs_one_chain=(K * args.steps / dtk) / (args.steps / dt))

    # K chains of this shape in ONE set of launches (dsm_batch_gibbs_update): the replicate chains of a G value
    batch = None
    if args.batch is None:
        args.batch = 4 if (dist is None and args.chains_per_gpu == 1) else 1
        batch_steps = 100                                        # its own length: the key is not the headline and must not rest on
    else:                                                        # a 2 ms call when the driver asks for --steps 20
        batch_steps = args.steps
    if args.batch > 1:
        K = args.batch
        ctxs = []
        for k in range(K):
            c2 = _lib.Context(dev)
            c2.set_counts(counts)
            c2.seed(2000 + k)
            c2.set_tau_rng(_lib.RNG_MT19937 if args.rng == "mt19937" else _lib.RNG_PHILOX)
            c2.set_state(tau_init, np.ascontiguousarray(gam.T), eta0)
            ctxs.append(c2)
        try:
            _lib.Context.batch_gibbs_update(ctxs, batch_steps)    # warm-up at the timed length: trace buffers are sized once
        except _lib.DesmanHipError:                              # shapes the aggregated mu/E pass does not cover (G > 16)
            for c2 in ctxs:
                c2.close()
            ctxs = []
    if args.batch > 1 and ctxs:
        one = ctxs[0]                                            # the same chain alone (its mu/E specification is the same in the batch)
        one.gibbs_update(batch_steps)
        t1s = []
        for _ in range(3):
            t0 = time.perf_counter(); one.gibbs_update(batch_steps); t1s.append(time.perf_counter() - t0)
        dt1 = float(np.median(t1s))
        tks = []
        for _ in range(3):                                       # median of three calls, like the headline
            t0 = time.perf_counter()
            _lib.Context.batch_gibbs_update(ctxs, batch_steps)
            tks.append(time.perf_counter() - t0)
        dtk = float(np.median(tks))
        for c2 in ctxs:
            c2.close()
        batch = dict(chains=K, ms_per_step_batch=1e3 * dtk / batch_steps, ms_per_step_per_chain=1e3 * dtk / batch_steps / K,
                     value=K * V * S * batch_steps / dtk, unit="V*S updates/s",
                     speedup_vs_one_chain_same_spec=(K * batch_steps / dtk) / (batch_steps / dt1),
                     speedup_vs_headline_chain=(K * batch_steps / dtk) / (args.steps / dt), steps=batch_steps)

    # per-kernel HIP-event timing (on the library's stream) for the roofline object
    ctx.sweep_stats(reset=True)
    ctx.set_timing(True)
    ctx.gibbs_update(20)
    tm = ctx.get_timing()
    ctx.set_timing(False)
    sw_steps, sw_exact = ctx.sweep_stats()           # wavefront-steps of those sweeps / left to the fp64 code by the fp32 screen
    k_us = {k: 1e3 * ms / max(n, 1) for k, (ms, n) in tm.items() if n}
    # What an event pair around ONE launch contains: timing mode drains the stream between launches, so every bracket holds the dispatch
    # latency of its launch on an idle queue as well as the kernel.  Measured here, not assumed: the main stream's brackets of those 20
    # iterations add up to more than 20 un-instrumented iterations take (the timed region above, where launches follow each other without a
    # gap: its iteration IS the sum of rocprofv3's kernel durations, profiles/r06_kernel_stats.csv) -- the excess per launch is that latency,
    # and a bracket less it is the kernel's own duration (it agrees with rocprofv3's begin -> end average to a few per cent; the raw
    # brackets are 3-5 us longer).  The refill of the uniforms runs on its own stream and is left out.
    timed_iters = max(1, tm.get("tau", (0.0, 0))[1])       # (the sweep runs once per iteration; the library's counters may cover more than this call)
    main = {k: v for k, v in tm.items() if v[1] and k in ("stats", "stats_big", "stats2", "stats_pat", "dirichlet", "tau", "finalize")}   # the iteration's launches on the main stream
    launches_per_iter = sum(n for _, n in main.values()) / float(timed_iters)
    events_us_per_iter = 1e3 * sum(ms for ms, _ in main.values()) / float(timed_iters)
    iter_us = 1e6 * float(np.median(rep_mine)) / args.steps
    dispatch_us = max(0.0, (events_us_per_iter - iter_us) / launches_per_iter) if launches_per_iter > 0 else 0.0
    spec = ctx.stats_spec()
    # algorithmic HBM bytes per launch: one pass over the int32 count tensor each,
    # plus the tau traffic (u8-equivalent: read for the mu/E pass; read + write + trace for the sweep)
    alg = {"tau": V * S * 16 + 3 * V * G, "stats": V * S * 16 + V * G}
    # candidate log-probability terms per sweep, SURVEY sec. 8(d): 16 V G S (+ 4 V S for the likelihood).  Each is evaluated by the
    # fp32 screening pass (hardware log2); the steps it cannot decide (tau_steps_fp64_frac) are re-evaluated in fp64, 12 of 16
    n_logs = 16.0 * V * G * S + 4.0 * V * S
    traffic, valu, valu_act = {}, {}, {}
    stats_kname = "stats_pat_kernel" if spec == 4 else "stats_agg_kernel" if spec >= 2 else "stats_kernel"
    tj, traffic_source = None, None
    pmc_error = None
    if args.pmc is None:                             # default: measure in this run when that is possible
        import shutil
        args.pmc = bool(world == 1 and shutil.which("rocprofv3") and not args.counts_npz)
    if not args.counts_npz:
        if args.pmc and rank == 0 and world == 1:
            try:
                tj = pmc_passes(V, S, G, args.depth_scale, save=args.pmc_save)
                traffic_source = ("measured by this run: three separate rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE x 2 per the "
                                  "gfx950 correction; WRITE_SIZE; the SQ VALU / MFMA group) of scripts/prof_gibbs.py at this shape")
            except Exception as e:                   # noqa: BLE001 -- a profiler failure must not lose the bench line
                pmc_error = "%s: %s" % (type(e).__name__, str(e)[-400:])
        if tj is None and os.path.exists(TRAFFIC_DB):
            tj = json.load(open(TRAFFIC_DB)).get(shape_key(V, S, G, args.depth_scale))
            traffic_source = ("profiles/pmc_traffic_by_shape.json[%s]: recorded by `bench.py --pmc` at this shape (separate rocprofv3 "
                              "--pmc passes: FETCH_SIZE x 2 per the gfx950 correction + WRITE_SIZE), not measured in this run"
                              % shape_key(V, S, G, args.depth_scale))
    if tj:
        for key, rec in tj.items():
            # the sweep + likelihood instantiation (tau_kernel<LPV, NSL, true, true>) and the stage-1 mu/E kernel
            name = "tau" if (key.startswith("void tau_kernel<") and key.endswith("true, true>")) else \
                   "stats" if (key.startswith(stats_kname) or key.startswith("void " + stats_kname)) else None
            if name:
                traffic[name] = rec.get("bytes_per_launch")
                valu[name] = rec.get("valu_insts")
                valu_act[name] = (rec.get("valu_active_cycles"), rec.get("sq_busy_cycles"))
    per_kernel = {}
    for name, kname in (("stats", stats_kname), ("tau", "tau_kernel")):
        us_raw = k_us.get(name, float("nan"))
        us = us_raw - dispatch_us                    # the kernel's own duration (see dispatch_us above)
        ach = alg[name] / (us * 1e-6) / 1e9
        per_kernel[kname] = dict(avg_kernel_us=us, avg_event_bracket_us=us_raw, algorithmic_bytes_per_launch=alg[name], achieved_GBps=ach,
                                 frac_of_8TBps=ach / 8000.0, traffic_bytes_pmc=traffic.get(name))
        if valu.get(name):
            per_kernel[kname].update(valu_insts_pmc=valu[name])
            act, busy = valu_act.get(name, (None, None))
            if act and busy:
                # what actually bounds the kernel, from the counters alone (no assumed cycles per instruction, no assumed clock):
                # SQ_ACTIVE_INST_VALU = quad-cycles (4 clocks) the wavefronts spent executing VALU instructions, summed over the device;
                # SQ_BUSY_CYCLES = clocks the SQ was busy, summed over its N_SE = 32 instances (8 XCDs x 4 shader engines) = 32 x the launch's
                # duration in shader clocks.  VALU-busy share of the 1024 SIMDs: active x 4 / (busy / 32 x 1024) = (active / busy) / 8.
                per_kernel[kname].update(valu_active_quadcycles_pmc=act, sq_busy_cycles_pmc=busy, valu_issue_frac=(act / busy) / 8.0,
                                         shader_clock_GHz_pmc=busy / 32.0 / (us * 1e-6) / 1e9)
            else:                                    # a record without the two counters (older pmc_traffic_by_shape.json): ~4 clocks per instruction at 2.4 GHz
                per_kernel[kname].update(valu_issue_frac=valu[name] * 4.0 / (1024 * us * 1e-6 * 2.4e9), valu_issue_frac_assumes="4 cycles per instruction at 2.4 GHz")
    launched, resident = ctx.tau_launch_info()
    rounds = launched / max(resident, 1)
    tau_launch = dict(launched_workgroups=launched, resident_workgroups=resident, rounds=rounds,
                      tail_frac=1.0 - rounds / float(np.ceil(rounds)) if rounds > 0 else None)
    dom = "stats" if k_us.get("stats", 0) >= k_us.get("tau", 0) else "tau"
    dk = per_kernel[(stats_kname if dom == "stats" else "tau_kernel")]
    roofline = dict(bound="hbm", bound_actual="valu_issue", kernel=(stats_kname if dom == "stats" else "tau_kernel"),
                    achieved=dk["achieved_GBps"], peak=8000.0, unit="GB/s",
                    frac=dk["frac_of_8TBps"], traffic=dk["traffic_bytes_pmc"],
                    traffic_source=(traffic_source if dk["traffic_bytes_pmc"] else None),
                    valu_issue_frac=dk.get("valu_issue_frac"),
                    avg_kernel_us=dk["avg_kernel_us"],
                    avg_event_bracket_us=dk["avg_event_bracket_us"],
                    dispatch_latency_us_per_launch=dispatch_us,
                    timing_note="avg_kernel_us = HIP-event bracket of a launch (timing mode drains the stream between launches) less the dispatch "
                                "latency measured in this run: (sum of the brackets of an iteration - the un-instrumented iteration of the timed "
                                "region) / launches per iteration; kernels_us are the raw brackets",
                    algorithmic_bytes_per_launch=dk["algorithmic_bytes_per_launch"],
                    per_kernel=per_kernel, log_terms_per_s_tau_kernel=n_logs / (k_us.get("tau", float("nan")) * 1e-6),
                    tau_steps_fp64_frac=(sw_exact / sw_steps) if sw_steps else None,
                    # workgroups a sweep launches / workgroups of tau_kernel resident at once: the launch runs as `rounds` waves of
                    # workgroups and the last, partly filled one is its tail (tail_frac = idle share of the slots of those rounds)
                    tau_launch=tau_launch,
                    stats_spec=spec,
                    note="both Gibbs kernels are VALU-issue bound, not HBM bound (bound_actual; per_kernel.valu_issue_frac, "
                         "PMC traffic ~ algorithmic bytes; profiles/, DESIGN.md sec. 3): the HBM fraction is reported "
                         "because the contract asks for it",
                    kernels_us=k_us)

    if rank == 0:
        its = args.steps / dt
        out = {
            "metric": "Gibbs iterations/sec (VxS updates/s) at V=%s S=%d G=%d" % (("%dk" % (V // 1000)) if V % 1000 == 0 else str(V), S, G),
            "value": world * V * S * its, "unit": "V*S updates/s",
            "gibbs_it_per_s_per_chain": its, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": data_label,
            "config": {"workload": "%s V=%d x S=%d, G=%d, full Gibbs iteration, 1 chain per GPU%s%s"
                                   % ("synthetic" if not args.counts_npz else "real", V, S, G, " (configs[2])" if (V, S, G) == (10000, 64, 8) else "",
                                      "" if args.depth_scale == 1.0 else ", read depth x%g" % args.depth_scale),
                       "V": V, "S": S, "G": G, "chains": world, "tau_rng": args.rng, "depth_scale": args.depth_scale},
            "ms_per_step_repeats": {"n": len(rep_dt), "median": 1e3 * dt / args.steps, "min": 1e3 * min(rep_dt) / args.steps,
                                    "max": 1e3 * max(rep_dt) / args.steps, "all": [1e3 * x / args.steps for x in rep_dt],
                                    "note": "each repeat = exactly `steps` iterations between barrier + synchronize, MAX over ranks; "
                                            "ms_per_step and value use the median repeat"},
            "ranks": sorted(ranks, key=lambda r: r["rank"]),
            "ms_per_step_per_rank": {"min": min(r["ms_per_step"] for r in ranks), "max": max(r["ms_per_step"] for r in ranks)},
            "launch": ("torch.distributed.run, %d rank(s), backend %s" % (world, "nccl (RCCL)" if backend == "nccl" else "gloo (rehearsal: ranks may share a device; not a benchmark configuration)")) if dist is not None else "single process, no process group",
            "roofline": roofline, "nmft": nmft,
            "fit_records": [dict(G=int(f[0]), seed=int(f[2]), lp_star=f[3], mean_dev=f[4]) for f in fits],
        }
        if pmc_error:
            out["pmc_error"] = pmc_error
        if nmft is not None:
            nsum = nmft_pmc_summary(tj, 1e3 * nmft["ms_per_iter"])
            if nsum:
                nmft["roofline"].update(traffic=nsum["traffic"], traffic_source=traffic_source,
                                        mfma_busy_frac=nsum["mfma_busy_frac"], mfma_insts_per_update=nsum["mfma_insts_per_update"],
                                        valu_insts_per_update=nsum["valu_insts_per_update"], kernels=nsum["kernels"])
        if multi:
            out["chains_per_gpu"] = multi
        if batch:
            out["batch"] = batch
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(S, G)
            out["speedup_vs_cpu_port"] = out["value"] / out["cpu_baseline"]["value"]
            out["speedup_vs_cpu_port_all_cores"] = out["value"] / out["cpu_baseline"]["all_cores"]["value"]
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
