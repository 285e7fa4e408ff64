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


def chain_cost(V, S, G, n_iter=None, nmf_updates=5000):
    """estimated cost of a chain in microseconds on one MI355X (only ratios matter to the scheduler).

    Fitted on the measurements of round 6 (profiles/r06_chain_cost_components.json: bench.py at V = 50k, S = 96, G = 2..12, and the
    config-3 line; `scripts/fit_chain_cost.py` makes the file): with kc = V S / 1000 (thousand cells)
      one Gibbs iteration   42 + kc (0.050 + 0.005 G)  [+ 18 + 1.2e-4 2^G S from G = 10: stage 2 of the mu/E pass as its own launch]
                            -- 0.498 / 0.627 ms at G = 9 / 12 there (round 5: 0.523 / 0.648), 0.100 ms at config 3; less where the mu/E
                            pass runs over tau words (large tables, 64 x 2^G <= V: a measured coefficient per G, below)
      one NMF update        12 + kc c(ceil(G / 4)), c = 0.01131 / 0.01467 / 0.01746 / 0.0188   -- 66 / 82 / 96 us at G <= 4 / <= 8 / <= 12 (K-blocks of four
                            haplotypes; round 5: 68 / 85 / 94, round 4: 85 / 96 / 107), 23 us at config 3
      host work per chain   0.23 s + 0.18 us per (position, haplotype): table filter, sampler set-up, the result files (Output_Results;
                            1.3 us before round 4 assembled the haplotype tables as byte fields)
    and a chain runs 2 n_iter iterations (burn-in + sampling, bin/desman:212-232) after up to 5000 NMF updates (Init_NMFT.py:98-115:
    the 1e-5 stop rarely fires).  n_iter = None: the cost of ONE iteration (chains of equal length compared).  Round 3's 6 + G had the
    slope and nothing else.  What no shape-only estimate can know is how well G fits the table: chains with too few haplotypes draw
    many more error reads, chains with too many run close races in the tau sweep (scripts/misfit_scan.py) -- up to twice the
    time (config-5 data, six strains: G = 2 / 3 take 1.61 / 1.71 s, G = 4 1.26 s).  Hence the default schedule is the work queue
    (WorkQueue below), which this estimate only orders, longest first."""
    kc = float(V) * float(S) / 1000.0
    gibbs = 42.0 + kc * (0.050 + 0.005 * G)
    if (G <= 2 and kc >= 500.0) or (G == 3 and kc >= 1000.0) or (4 <= G <= 8 and kc >= 2500.0 and 64 * 2 ** int(G) <= V):
        # the mu/E pass over tau words (kernels_stats.hip: stats_spec, spec 4) where few words cover many positions: a measured
        # coefficient per G (V = 50k, S = 96, profiles/r06_chain_cost_components.json)
        gibbs = 42.0 + kc * PAT_COEF.get(int(G), 0.050 + 0.005 * G)
    if G >= 10:
        gibbs += 18.0 + 1.2e-4 * float(1 << min(int(G), 30)) * float(S)
    if n_iter is None:
        return gibbs
    nmf = 12.0 + kc * NMF_COEF.get((int(G) + 3) // 4, 0.0188 + 0.0015 * ((int(G) + 3) // 4 - 4))
    host = 0.23e6 + 0.18 * float(V) * float(G)
    return 2.0 * float(n_iter) * gibbs + float(nmf_updates) * nmf + host


# us per thousand cells and Gibbs iteration where the mu/E pass runs over tau words (chain_cost), G -> coefficient
PAT_COEF = {1: 0.022, 2: 0.02859, 3: 0.03858, 4: 0.04928, 5: 0.05843, 6: 0.05984, 7: 0.07439, 8: 0.09558}
# us per thousand cells and NMF update, ceil(G / 4) -> coefficient
NMF_COEF = {1: 0.01131, 2: 0.01467, 3: 0.01746, 4: 0.0188}


class WorkQueue:
    """Units of work handed out one at a time to whoever asks next -- across the ranks of ONE node (the only topology the path
    has: scripts/runDesman.sh:15-21 starts its jobs on one machine).  A rank that drew short chains simply comes back sooner, so
    the makespan does not hang on the cost estimate (it only orders the queue, longest first).  Three carriers of the counter:
    a lock in this process (one rank), a small file under flock() (ranks started by desman_amd.launch.spawn_ranks, which makes the
    file and passes its path in DESMAN_SWEEP_QUEUE), the rendezvous store of torch.distributed (`store.add`, ranks started by
    torch.distributed.run).  next() returns 0, 1, 2, ... exactly once each over all callers."""

    def __init__(self, file_path=None, store=None, key="desman_sweep_queue"):
        import threading
        self._lock = threading.Lock()
        self._n = 0
        self._epoch = -1
        self._path, self._store, self._key = file_path, store, key

    def begin(self):
        """a new list of units (one per run_chains call; every rank makes the same calls in the same order): the counter
        starts at 0 again -- its own 8 bytes of the file / its own key of the store, so a rank that is already in call k + 1
        never takes numbers from call k's counter"""
        with self._lock:
            self._epoch += 1
            self._n = 0

    def next(self):
        epoch = max(self._epoch, 0)
        if self._store is not None:
            return int(self._store.add("%s/%d" % (self._key, epoch), 1)) - 1
        if self._path is not None:
            import fcntl
            import struct
            with self._lock, open(self._path, "r+b") as f:      # (the lock: flock is per open file description, threads share none here)
                fcntl.flock(f, fcntl.LOCK_EX)
                f.seek(8 * epoch)
                raw = f.read(8)
                n = struct.unpack("<q", raw)[0] if len(raw) == 8 else 0
                f.seek(8 * epoch)
                f.write(struct.pack("<q", n + 1))
                f.flush()
                os.fsync(f.fileno())
                fcntl.flock(f, fcntl.LOCK_UN)
            return n
        with self._lock:
            n = self._n
            self._n += 1
        return n


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


def sweep_specs(g_values, n_reps, V, S, n_iter=None):
    """(G, seed) grid of a model-selection sweep: seeds 0..n_reps-1 per G (runDesman.sh:15-19); n_iter = the chains' -i (the whole
    chain is then costed: NMF start + 2 n_iter iterations + result files)."""
    specs = []
    for G in g_values:
        for r in range(n_reps):
            specs.append(dict(chain=len(specs), G=int(G), seed=int(r), cost=chain_cost(V, S, G, n_iter)))
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


class WorkerError(RuntimeError):
    """a worker thread stopped outside a chain (the queue's flock / store failed ...): raised by run_chains on every rank after the gather;
    .records holds every chain's record (the chains that ran nowhere carry failed = 1)"""

    def __init__(self, msg, records):
        super().__init__(msg)
        self.records = records


def run_chains(specs, run_fn, dist=None, device=None, concurrency=1, batch_fn=None, batch=1, comm=None, queue=None):
    """Run `run_fn(spec) -> dict(REC_FIELDS...)` for this rank's share of `specs` and gather
    every chain's record on all ranks.  `dist` = an initialised torch.distributed module
    (or None for a single process).  `concurrency` chains of a rank run at the same time in
    threads (each on its own context / HIP streams; ctypes releases the GIL while a launch
    sequence is in flight): small-V chains are latency-bound, so several of them share a GPU
    well.  Returns the records sorted by chain id.

    Which rank runs which chain: `queue` = None -- the static plan, longest-processing-time-first over the cost estimates
    (lpt_assign; the same on every rank, nothing exchanged); `queue` = a WorkQueue shared by the ranks -- every worker thread
    of every rank takes the next unit of the list (sorted by decreasing estimate) when it is free.  A chain's result does not
    depend on where it ran (its seeds are its own), so both schedules write the same files.

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
    batched = batch_fn is not None and batch > 1
    units = group_units(specs, batch if batch_fn is not None else 1)
    ucost = [sum(specs[i]["cost"] for i in u) for u in units]
    unit_bins = lpt_assign(ucost, world)
    bins = [[i for ui in ub for i in units[ui]] for ub in unit_bins]          # chain ids per rank, unit by unit (static plan)

    import logging
    import threading
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

    def unit(ui):
        """[(chain id, record or exception)] of one unit: a batched run of its chains where that applies (a unit whose batched run
        raises falls back to its chains one by one, and those to the re-queue / failed-record rule below), else chain by chain"""
        ids = units[ui]
        if batched and len(ids) > 1:
            t0 = time.perf_counter()
            try:
                recs = [dict(r) for r in batch_fn([specs[i] for i in ids])]
                if len(recs) != len(ids):
                    raise RuntimeError("batch_fn returned %d records for %d chains" % (len(recs), len(ids)))
            except (Exception, SystemExit) as e:             # noqa: BLE001
                log.warning("batched unit %s failed on rank %d (%s: %s): its chains run one by one", ids, rank, type(e).__name__, e)
                recs = None
            if recs is not None:
                out = []
                for cid, rec in zip(ids, recs):
                    rec.setdefault("wall_s", (time.perf_counter() - t0) / len(ids))
                    rec["chain"] = cid
                    rec.setdefault("failed", 0.0)
                    out.append((cid, [float(rec[k]) for k in REC_FIELDS]))
                return out
        return [(cid, one(cid)) for cid in ids]

    # where this rank's next unit comes from
    take_lock = threading.Lock()
    if queue is None:
        pending = list(unit_bins[rank])                      # LPT order within the rank: longest units start first
        n_mine = len(pending)

        def take():
            with take_lock:
                return pending.pop(0) if pending else None
    else:
        order = sorted(range(len(units)), key=lambda u: (-ucost[u], u))      # the same list on every rank
        n_mine = len(order)                                  # (an upper bound: what this rank could end up running)
        queue.begin()

        def take():
            k = queue.next()
            return order[k] if k < len(order) else None

    done = []                                                # (chain id, record or exception), in order of completion
    worker_errors = []                                       # what a worker thread died of outside a chain (the queue's carrier)

    def worker():
        try:
            while True:
                ui = take()
                if ui is None:
                    return
                res = unit(ui)
                with take_lock:
                    done.extend(res)
        except KeyboardInterrupt:                            # (only ever in the main thread: the user's Ctrl-C is not held back until after the gather)
            raise
        except BaseException as e:                           # noqa: BLE001 -- a thread must not die silently: see after the gather
            log.error("worker thread of rank %d stopped: %s: %s", rank, type(e).__name__, e)
            with take_lock:
                worker_errors.append(e)

    n_workers = max(1, min(int(concurrency), n_mine))
    if n_workers > 1:
        if not batched:
            os.environ.setdefault("DESMAN_HIP_NMFT_GRAPH", "1")  # replayed NMFT batches: see api.hip (dsm_nmft_factorize)
        # more than four live hardware queues stretch every small kernel to ~55 us: the MT19937 refill of
        # a chain then runs on the chain's own stream (35-chain sweep at V = 1000, 8 at a time: 4.2 -> 2.6 s)
        os.environ.setdefault("DESMAN_HIP_ONE_STREAM", "1")
        threads = [threading.Thread(target=worker, name="desman-chain-%d" % k) for k in range(n_workers)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    else:
        worker()
    mine = []
    for cid, res in sorted(done, key=lambda x: x[0]):
        if isinstance(res, BaseException):                   # second and last attempt, alone on the device
            res = one(cid)
            if isinstance(res, BaseException):
                res = failed_record(cid)
        mine.append(res)
    width = (max(len(b) for b in bins) if queue is None else len(specs)) if specs else 0
    # (one more row than records: its first word says whether a worker thread of this rank stopped outside a chain -- the ranks
    # learn it from each other in the one gather and take the same path afterwards)
    buf = np.full((max(width, 1) + 1, len(REC_FIELDS)), np.nan)
    if mine:
        buf[:len(mine)] = np.array(mine)
    buf[-1, 1] = 1.0 if worker_errors else 0.0
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
    flags = rows.reshape(world, -1, len(REC_FIELDS))[:, -1, 1] if world > 1 or comm is not None or dist is not None else rows[-1:, 1]
    bad_ranks = [int(r) for r in np.nonzero(np.nan_to_num(flags) > 0)[0]]
    rows = rows[~np.isnan(rows[:, 0])]
    rows = rows[np.argsort(rows[:, 0])]
    # every chain exactly once -- after the gather, so that every rank reaches the collective and every rank sees the same hole.
    # A unit a dead worker thread had taken (flock / store error in take()) is gone from the queue: its chains get failed
    # records, model selection skips them, and the error is raised where it happened.
    got = [int(r[0]) for r in rows]
    if len(set(got)) != len(got):
        raise RuntimeError("run_chains: chains gathered twice: %s" % sorted(c for c in set(got) if got.count(c) > 1))
    missing = sorted(set(range(len(specs))) - set(got))
    if missing:
        log.error("run_chains: %d chain(s) ran nowhere (%s): failed records", len(missing), missing[:16])
        rows = np.concatenate([rows, np.array([failed_record(c) for c in missing])], axis=0)
        rows = rows[np.argsort(rows[:, 0])]
    recs = [dict(zip(REC_FIELDS, r.tolist())) for r in rows]
    if bad_ranks:
        # raised on EVERY rank (round 5 raised on the rank it happened on only: with that rank being 0, no Dev.csv was written although
        # the failed records had just been built for it), with the records attached
        err = WorkerError("run_chains: worker thread(s) of rank(s) %s stopped outside a chain" % bad_ranks, recs)
        if worker_errors:
            raise err from worker_errors[0]
        raise err
    return recs


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
    ap.add_argument("--schedule", choices=["queue", "plan"], default="queue", help="queue (default): every rank takes the next chain of "
                    "one list, longest estimate first, when it is free (a counter file under flock / the torch rendezvous store) -- chains "
                    "that fit the table badly take up to twice their estimate; plan: the static longest-processing-time-first assignment")
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
    dev = local                                              # the device this rank's chains run on
    if under_launcher and args.comm == "torch":
        import torch
        import torch.distributed as dist
        backend = launch.dist_backend()                      # nccl (RCCL) unless DESMAN_DIST_BACKEND=gloo: the rehearsal with ranks sharing a device
        dev = launch.bind_device(local, torch.cuda.device_count(), "desman-sweep")
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
            dev_t = torch.device("cuda", dev)
        else:
            dist.init_process_group("gloo")                  # (the gather's tensors stay on the host: run_chains, device=None)
        if dist.get_world_size() != args.gpus:
            launch._die("desman-sweep", "process group of %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    elif under_launcher:
        from .comm import Comm
        comm = Comm.from_env(device=local)
        if comm.world != args.gpus:
            launch._die("desman-sweep", "communicator of %d ranks, --gpus %d" % (comm.world, args.gpus))
    frame = p.read_csv(args.variant_file, header=0, index_col=0)
    V, S = frame.shape[0], (frame.shape[1] - 1) // 4
    specs = sweep_specs(range(args.gmin, args.gmax + 1), args.reps, V, S, n_iter=args.no_iter)
    queue = queue_made = None
    if args.schedule == "queue" and world > 1:
        if comm is not None:
            qpath = os.environ.get("DESMAN_SWEEP_QUEUE")
            if not qpath:
                # ranks started by another launcher (python -m torch.distributed.run -m desman_amd.chains ...): no counter file was made
                # for us.  Rank 0 makes one (mkstemp: a fresh name, O_EXCL, mode 0600 -- round 5's name was predictable from the
                # rendezvous and opened with "wb", which follows a symlink) and the ranks learn the path and rank 0's host through the
                # communicator: the counter is a file under flock(), so the ranks must share a file system -- one node -- and a rank
                # on another host stops here with that message instead of dying in take() after the work is done.
                import socket
                import tempfile
                host = socket.gethostname().encode()[:255]
                msg = np.zeros(1024, np.float64)
                if comm.rank == 0:
                    fd, made = tempfile.mkstemp(prefix="desman_queue_")
                    os.write(fd, b"\0" * 8)
                    os.close(fd)
                    queue_made = made
                    pb = made.encode()
                    msg[0], msg[1] = len(pb), len(host)
                    msg[2:2 + len(pb)] = np.frombuffer(pb, np.uint8)
                    msg[514:514 + len(host)] = np.frombuffer(host, np.uint8)
                got = comm.allgather(msg).reshape(comm.world, -1)[0]
                qpath = bytes(got[2:2 + int(got[0])].astype(np.uint8)).decode()
                host0 = bytes(got[514:514 + int(got[1])].astype(np.uint8))
                if host0 != host or not os.path.exists(qpath):
                    launch._die("desman-sweep", "rank %d runs on host %r, rank 0 on %r: the work queue's counter is a file under flock() -- "
                                "one node; use --schedule lpt across nodes" % (comm.rank, host.decode(), host0.decode()))
            queue = WorkQueue(file_path=qpath)
        else:
            from torch.distributed.distributed_c10d import _get_default_store
            queue = WorkQueue(store=_get_default_store())
    extra = ["-m", str(args.min_coverage)] + (["-r", str(args.random_select)] if args.random_select else [])
    runner = gibbs_chain_runner(args.variant_file, args.no_iter, dev, args.output_stub, extra)
    failed_run = None
    try:
        recs = run_chains(specs, runner, dist, device=dev_t, concurrency=args.concurrency, batch_fn=runner.batch if args.batch > 1 else None,
                          batch=min(args.batch, 8), comm=comm, queue=queue)
        if (comm.rank if comm is not None else (0 if dist is None else dist.get_rank())) == 0:
            write_dev_csv(args.output_stub + "_Dev.csv", recs)
            print(json.dumps(recs))
            from . import resolvenhap                      # f2: posterior-deviance model selection over the sweep
            resolvenhap.resolve(args.output_stub)
    except WorkerError as e:
        # a worker thread of SOME rank stopped outside a chain: every rank learnt it in the gather (run_chains) and every rank takes this
        # path -- rank 0 still writes Dev.csv and runs model selection on the records (the lost chains carry failed = 1), then all
        # exit non-zero
        failed_run = e
        if (comm.rank if comm is not None else (0 if dist is None else dist.get_rank())) == 0:
            write_dev_csv(args.output_stub + "_Dev.csv", e.records)
            print(json.dumps(e.records))
            from . import resolvenhap
            resolvenhap.resolve(args.output_stub)
    finally:
        # the same tidy-up on every rank and every path
        if dist is not None:
            try:
                dist.destroy_process_group()
            except Exception:                              # noqa: BLE001 -- the group may be gone with a dead peer
                pass
        if comm is not None:
            if queue_made and comm.rank == 0:              # (after the gather: nobody takes numbers any more)
                try:
                    os.unlink(queue_made)
                except OSError:
                    pass
            comm.close()
    if failed_run is not None:
        raise SystemExit("desman-sweep: %s" % failed_run)


if __name__ == "__main__":
    main(sys.argv[1:])
