"""Multi-GPU chain scheduler: independent Gibbs chains (seed replicates x G
values) are the only parallel axis the reference has -- it spawns one process
per (G, replicate) (scripts/runDesman.sh:15-21, complete_example/README.md:
611-627) and "gathers" by concatenating fit.txt files.  Here: one process per
GPU (torch.distributed; backend "nccl" = RCCL over xGMI, "gloo" on CPU for
tests), chains assigned by longest-processing-time-first, NO collective on the
data path, and one all_gather of a fixed-size fit record per chain at the end.

Launch:  python -m desman_amd.chains --gpus N freq.csv ...      (starts its own N ranks: desman_amd/launch.py)
   or:   python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
             --master-addr 127.0.0.1 --master-port P -m desman_amd.chains [--gpus N] freq.csv ...
With --gpus the world must be exactly N ranks (exit status 2 otherwise); without it the launcher's world is taken as it is.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REC_FIELDS = ("chain", "G", "seed", "G_final", "lp_star", "mean_dev", "iters", "wall_s", "failed")


def chain_cost(V, S, G):
    """relative cost of one Gibbs iteration, as measured on MI355X (profiles/r02_shape_scan.txt: V = 50k, S = 96 takes
    0.32 / 0.41 / 0.45 / 0.73 ms at G = 2 / 4 / 5 / 12, i.e. ~0.25 + 0.04 G): the mu/E pass costs the same per cell
    whatever G is, the tau sweep grows with G.  (Round 1's 4 G + G^2 over-weighted the large-G chains 7-fold.)"""
    return float(V) * float(S) * (6.0 + float(G))


def lpt_assign(costs, n_ranks):
    """longest-processing-time-first greedy bin packing -> list of chain ids per rank
    (deterministic: ties broken by chain id, then by rank)."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    load = [0.0] * n_ranks
    bins = [[] for _ in range(n_ranks)]
    for i in order:
        r = min(range(n_ranks), key=lambda k: (load[k], k))
        bins[r].append(i)
        load[r] += costs[i]
    return bins


def sweep_specs(g_values, n_reps, V, S):
    """(G, seed) grid of a model-selection sweep: seeds 0..n_reps-1 per G (runDesman.sh:15-19)."""
    specs = []
    for G in g_values:
        for r in range(n_reps):
            specs.append(dict(chain=len(specs), G=int(G), seed=int(r), cost=chain_cost(V, S, G)))
    return specs


def group_units(specs, batch):
    """chain ids grouped into the units that are scheduled together: with batch > 1 the replicate chains of a G value, at
    most `batch` per unit (they share every kernel launch of the Gibbs loop: dsm_batch_gibbs_update), else one chain each"""
    if batch <= 1:
        return [[i] for i in range(len(specs))]
    by_g = {}
    for i, sp in enumerate(specs):
        by_g.setdefault(sp.get("G"), []).append(i)
    units = []
    for g in sorted(by_g, key=lambda x: (x is None, x)):
        ids = by_g[g]
        units += [ids[j:j + batch] for j in range(0, len(ids), batch)]
    return units


def run_chains(specs, run_fn, dist=None, device=None, concurrency=1, batch_fn=None, batch=1, comm=None):
    """Run `run_fn(spec) -> dict(REC_FIELDS...)` for this rank's share of `specs` and gather
    every chain's record on all ranks.  `dist` = an initialised torch.distributed module
    (or None for a single process).  `concurrency` chains of a rank run at the same time in
    threads (each on its own context / HIP streams; ctypes releases the GIL while a launch
    sequence is in flight): small-V chains are latency-bound, so several of them share a GPU
    well.  Returns the records sorted by chain id.

    Failure handling ("a failed GPU's chains are simply re-queued"): a chain whose run_fn raises
    is re-queued ONCE on the same rank after the rank's other chains; if it fails again its record carries
    failed = 1 and NaN fit values (model selection skips it).  A rank therefore always reaches the all_gather:
    one bad chain (bad input for that G, an out-of-memory context, a device error) never strands its peers
    in the collective."""
    # the gather goes through `comm` (desman_amd.comm.Comm: the library's own RCCL communicator, no torch) or through `dist`
    # (an initialised torch.distributed: "nccl" = RCCL on GPUs, "gloo" in the CPU tests); neither: one process
    if comm is not None:
        world, rank = comm.world, comm.rank
    else:
        world = dist.get_world_size() if dist is not None else 1
        rank = dist.get_rank() if dist is not None else 0
    units = group_units(specs, batch if batch_fn is not None else 1)
    unit_bins = lpt_assign([sum(specs[i]["cost"] for i in u) for u in units], world)
    bins = [[i for ui in ub for i in units[ui]] for ub in unit_bins]          # chain ids per rank, unit by unit

    import logging
    log = logging.getLogger("desman_amd.chains")

    def one(cid):
        """record of chain cid, or the exception it raised (never propagates: see the docstring)"""
        t0 = time.perf_counter()
        try:
            rec = dict(run_fn(specs[cid]))
        except (Exception, SystemExit) as e:                 # noqa: BLE001 -- any failure of one chain is contained (the CLI
            #                                                  reports bad input through sys.exit: a BaseException)
            log.warning("chain %d (G=%s, seed=%s) failed on rank %d: %s: %s", cid, specs[cid].get("G"), specs[cid].get("seed"),
                        rank, type(e).__name__, e)
            return e
        rec.setdefault("wall_s", time.perf_counter() - t0)
        rec["chain"] = cid
        rec.setdefault("failed", 0.0)
        return [float(rec[k]) for k in REC_FIELDS]

    def failed_record(cid):
        sp = specs[cid]
        rec = dict(chain=cid, G=sp.get("G", np.nan), seed=sp.get("seed", np.nan), G_final=np.nan, lp_star=np.nan,
                   mean_dev=np.nan, iters=0, wall_s=np.nan, failed=1.0)
        return [float(rec[k]) for k in REC_FIELDS]

    if batch_fn is not None and batch > 1:
        # the units of this rank one after the other, each as one batched run; a unit whose batched run raises falls back to
        # its chains one by one (and those to the re-queue / failed-record rule below)
        def unit(ui):
            ids = units[ui]
            out = []
            recs = None
            if len(ids) > 1:
                t0 = time.perf_counter()
                try:
                    recs = [dict(r) for r in batch_fn([specs[i] for i in ids])]
                    if len(recs) != len(ids):
                        raise RuntimeError("batch_fn returned %d records for %d chains" % (len(recs), len(ids)))
                except (Exception, SystemExit) as e:         # noqa: BLE001
                    log.warning("batched unit %s failed on rank %d (%s: %s): its chains run one by one", ids, rank, type(e).__name__, e)
                    recs = None
                if recs is not None:
                    for cid, rec in zip(ids, recs):
                        rec.setdefault("wall_s", (time.perf_counter() - t0) / len(ids))
                        rec["chain"] = cid
                        rec.setdefault("failed", 0.0)
                        out.append([float(rec[k]) for k in REC_FIELDS])
            if recs is None:
                out += [one(cid) for cid in ids]
            return out
        if concurrency > 1 and len(unit_bins[rank]) > 1:         # units (different G) side by side: their host work overlaps
            from concurrent.futures import ThreadPoolExecutor
            os.environ.setdefault("DESMAN_HIP_ONE_STREAM", "1")  # one hardware queue per chain (see below)
            with ThreadPoolExecutor(max_workers=concurrency) as pool:
                first = [r for rs in pool.map(unit, unit_bins[rank]) for r in rs]
        else:
            first = [r for ui in unit_bins[rank] for r in unit(ui)]
    elif concurrency > 1 and len(bins[rank]) > 1:
        from concurrent.futures import ThreadPoolExecutor
        os.environ.setdefault("DESMAN_HIP_NMFT_GRAPH", "1")      # replayed NMFT batches: see api.hip (dsm_nmft_factorize)
        # more than four live hardware queues stretch every small kernel to ~55 us: the MT19937 refill of
        # a chain then runs on the chain's own stream (35-chain sweep at V = 1000, 8 at a time: 4.2 -> 2.6 s)
        os.environ.setdefault("DESMAN_HIP_ONE_STREAM", "1")
        with ThreadPoolExecutor(max_workers=concurrency) as pool:
            first = list(pool.map(one, bins[rank]))          # LPT order: longest chains start first
    else:
        first = [one(cid) for cid in bins[rank]]
    mine = []
    for cid, res in zip(bins[rank], first):
        if isinstance(res, BaseException):                   # second and last attempt, alone on the device
            res = one(cid)
            if isinstance(res, BaseException):
                res = failed_record(cid)
        mine.append(res)
    width = max(len(b) for b in bins) if specs else 0
    buf = np.full((max(width, 1), len(REC_FIELDS)), np.nan)
    if mine:
        buf[:len(mine)] = np.array(mine)
    if comm is not None:
        rows = comm.allgather(buf.reshape(-1)).reshape(-1, len(REC_FIELDS))      # the path's single exchange step
    elif dist is None:
        rows = buf
    else:
        import torch
        t = torch.from_numpy(buf).to(device if device is not None else "cpu")
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)                         # the path's single exchange step
        rows = np.concatenate([o.cpu().numpy() for o in out], axis=0)
    rows = rows[~np.isnan(rows[:, 0])]
    rows = rows[np.argsort(rows[:, 0])]
    return [dict(zip(REC_FIELDS, r.tolist())) for r in rows]


class _ThreadLogRouter(__import__("logging").Handler):
    """routes log records to the log_file.txt of the chain the emitting thread is running (the
    reference has one process, hence one log file, per chain)."""

    def __init__(self):
        super().__init__()
        import threading
        self._files, self._lock = {}, threading.Lock()
        self.setFormatter(__import__("logging").Formatter('%(asctime)s:%(levelname)s:%(name)s:%(message)s'))

    def open(self, path):
        import threading
        with self._lock:
            self._files[threading.get_ident()] = open(path, "w")

    def switch(self, handle):
        """make an open file the current one of this thread (a batch of chains run by one thread)"""
        import threading
        with self._lock:
            self._files[threading.get_ident()] = handle

    def close_current(self):
        import threading
        with self._lock:
            f = self._files.pop(threading.get_ident(), None)
        if f:
            f.close()

    def emit(self, record):
        f = self._files.get(record.thread)
        if f is not None:
            f.write(self.format(record) + "\n")
            f.flush()


def gibbs_chain_runner(variant_file, n_iter, device, out_stub, extra_args=()):
    """run_fn for real chains on one GPU: the whole `desman` run for (G, seed) -- NMFT init,
    burn-in, removeDegenerate, sampling, all output files -- into `<stub>_<G>_<seed>/`, the
    directory layout scripts/runDesman.sh:15-21 produces and scripts/resolvenhap.py reads.
    Thread-safe: every chain has its own context, its own logical GSL stream and its own log file."""
    import logging

    from . import _lib, cli, sampletau

    _lib.load()                                           # before any worker thread
    router = _ThreadLogRouter()
    for h in list(logging.root.handlers):
        logging.root.removeHandler(h)
    logging.root.addHandler(router)
    logging.root.setLevel(logging.INFO)

    def run(spec):
        t0 = time.perf_counter()
        G, seed = spec["G"], spec["seed"]
        d = "%s_%d_%d" % (out_stub, G, seed)
        os.makedirs(d, exist_ok=True)
        sampletau.use_thread_local_rng(True)
        router.open(os.path.join(d, "log_file.txt"))
        try:
            cli.main([variant_file, "-g", str(G), "-s", str(seed), "-i", str(n_iter), "-o", d, "--device", str(device)]
                     + list(extra_args))
        finally:
            router.close_current()
            sampletau.use_thread_local_rng(False)         # the calling thread gets the process-global stream back
        _, gt, ht, lp, dev = open(os.path.join(d, "fit.txt")).read().strip().split(",")
        return dict(G=G, seed=seed, G_final=int(ht), lp_star=float(lp), mean_dev=float(dev), iters=2 * n_iter,
                    wall_s=time.perf_counter() - t0)

    def run_batch(group):
        """the same for the replicate chains of one G value, their Gibbs iterations batched (cli.main_replicates)"""
        dirs, argvs, logs = [], [], []
        for spec in group:
            d = "%s_%d_%d" % (out_stub, spec["G"], spec["seed"])
            os.makedirs(d, exist_ok=True)
            dirs.append(d)
            argvs.append([variant_file, "-g", str(spec["G"]), "-s", str(spec["seed"]), "-i", str(n_iter), "-o", d,
                          "--device", str(device)] + list(extra_args))
            logs.append(open(os.path.join(d, "log_file.txt"), "w"))
        sampletau.use_thread_local_rng(True)
        try:
            cli.main_replicates(argvs, on_chain=lambda k: router.switch(logs[k]))
        finally:
            router.close_current()
            for f in logs:
                if not f.closed:
                    f.close()
            sampletau.use_thread_local_rng(False)
        recs = []
        for spec, d in zip(group, dirs):
            _, gt, ht, lp, dev = open(os.path.join(d, "fit.txt")).read().strip().split(",")
            recs.append(dict(G=spec["G"], seed=spec["seed"], G_final=int(ht), lp_star=float(lp), mean_dev=float(dev), iters=2 * n_iter))
        return recs
    run.batch = run_batch
    return run


def write_dev_csv(path, records):
    """`H,G,LP,Dev` table the reference builds with `cat */fit.txt` (complete_example/README.md:626-627)."""
    with open(path, "w") as f:
        f.write("H,G,LP,Dev\n")
        for r in records:
            if r.get("failed"):
                continue                                   # a chain that failed twice has no fit
            f.write("%d,%d,%f,%f\n" % (r["G"], r["G_final"], r["lp_star"], r["mean_dev"]))


def main(argv=None):
    ap = argparse.ArgumentParser(prog="desman-sweep", description="G-sweep of independent Gibbs chains over the "
                                 "GPUs of one node (one process per GPU)")
    ap.add_argument("variant_file")
    ap.add_argument("--gmin", type=int, default=2)
    ap.add_argument("--gmax", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("-i", "--no_iter", type=int, default=100)
    ap.add_argument("-m", "--min_coverage", type=float, default=5.0)
    ap.add_argument("-r", "--random_select", type=int, default=None)
    ap.add_argument("-o", "--output_stub", default="sweep")
    ap.add_argument("-c", "--concurrency", type=int, default=4, help="chains running at the same time per GPU")
    ap.add_argument("-b", "--batch", type=int, default=1, help="replicate chains of a G value (up to 8) share every kernel "
                    "launch of the Gibbs loop instead of running as separate chains (small tables: several times the throughput)")
    ap.add_argument("--gpus", type=int, default=None, help="GPUs of this node to spread the chains over (one process each). "
                    "A plain `desman-sweep --gpus N` starts its own N ranks; under torch.distributed.run the world must be N")
    ap.add_argument("--comm", choices=["rccl", "torch"], default="rccl", help="who carries the final gather of the fit records: rccl = "
                    "the library's own RCCL communicator (include/desman_hip.h: dsm_comm_*; no torch in the process, ranks started by "
                    "desman_amd.launch.spawn_ranks), torch = torch.distributed with backend nccl (ranks started by torch.distributed.run)")
    args = ap.parse_args(argv)
    from . import launch
    if args.gpus is None:                                     # the launcher's world as it is (1 for a plain process)
        args.gpus = int(os.environ.get("WORLD_SIZE", "1")) if "RANK" in os.environ else 1
    _, local, world, under_launcher = launch.ensure_world(args.gpus, sys.argv[1:] if argv is None else list(argv),
                                                          module="desman_amd.chains", prog="desman-sweep",
                                                          launcher="spawn" if args.comm == "rccl" else "torchrun")
    import pandas as p
    dist = comm = None
    dev_t = None
    if under_launcher and args.comm == "torch":
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        dev_t = torch.device("cuda", local)
        if dist.get_world_size() != args.gpus:
            launch._die("desman-sweep", "process group of %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    elif under_launcher:
        from .comm import Comm
        comm = Comm.from_env(device=local)
        if comm.world != args.gpus:
            launch._die("desman-sweep", "communicator of %d ranks, --gpus %d" % (comm.world, args.gpus))
    frame = p.read_csv(args.variant_file, header=0, index_col=0)
    V, S = frame.shape[0], (frame.shape[1] - 1) // 4
    specs = sweep_specs(range(args.gmin, args.gmax + 1), args.reps, V, S)
    extra = ["-m", str(args.min_coverage)] + (["-r", str(args.random_select)] if args.random_select else [])
    if args.batch > 1:
        # batched units draw mu/E from the aggregated specification; chains of this sweep that run one by one (a unit of one
        # chain, the fallback of a failed unit) follow it too, so that a (G, seed) gives the same draws whatever -b is
        os.environ["DESMAN_HIP_STATS_SPEC"] = "2"
    runner = gibbs_chain_runner(args.variant_file, args.no_iter, local, args.output_stub, extra)
    recs = run_chains(specs, runner, dist, device=dev_t, concurrency=args.concurrency, batch_fn=runner.batch if args.batch > 1 else None,
                      batch=min(args.batch, 8), comm=comm)
    if (comm.rank if comm is not None else (0 if dist is None else dist.get_rank())) == 0:
        write_dev_csv(args.output_stub + "_Dev.csv", recs)
        print(json.dumps(recs))
        from . import resolvenhap                      # f2: posterior-deviance model selection over the sweep
        resolvenhap.resolve(args.output_stub)
    if dist is not None:
        dist.destroy_process_group()
    if comm is not None:
        comm.close()


if __name__ == "__main__":
    main(sys.argv[1:])
