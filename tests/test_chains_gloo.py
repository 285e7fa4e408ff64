"""Multi-process (world_size 2, gloo, CPU) tests of the chain scheduler and the
fit-record gather -- the N>1 path of bench.py / desman_amd.chains."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from desman_amd import chains

HERE = os.path.dirname(os.path.abspath(__file__))


def test_lpt_assignment_properties():
    costs = [chains.chain_cost(50000, 96, g) for g in range(2, 13) for _ in range(5)]      # config 5: 55 chains
    bins = chains.lpt_assign(costs, 8)
    assert sorted(i for b in bins for i in b) == list(range(55))
    load = [sum(costs[i] for i in b) for b in bins]
    assert max(load) <= sum(costs) / 8 + max(costs)                     # LPT bound
    assert max(load) / (sum(costs) / 8) < 1.15
    assert chains.lpt_assign(costs, 8) == bins                           # deterministic
    assert chains.lpt_assign([3.0, 1.0], 4) == [[0], [1], [], []]


def test_single_process_equals_serial():
    specs = chains.sweep_specs([2, 3], 2, 100, 8)
    recs = chains.run_chains(specs, lambda s: dict(G=s["G"], seed=s["seed"], G_final=s["G"], lp_star=-1.0 * s["G"],
                                                   mean_dev=2.0, iters=1, wall_s=0.0))
    assert [(int(r["G"]), int(r["seed"])) for r in recs] == [(2, 0), (2, 1), (3, 0), (3, 1)]
    assert [int(r["chain"]) for r in recs] == [0, 1, 2, 3]


def test_two_rank_gloo_gather(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29653", os.path.join(HERE, "_gloo_worker.py"), str(tmp_path)]
    subprocess.run(cmd, check=True, env=env, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r0 = json.load(open(tmp_path / "rank0.json"))
    r1 = json.load(open(tmp_path / "rank1.json"))
    assert r0["recs"] == r1["recs"]                                      # all_gather: every rank has every record
    assert sorted(r0["mine"] + r1["mine"]) == list(range(15)) and not set(r0["mine"]) & set(r1["mine"])
    specs = chains.sweep_specs(range(2, 7), 3, V=1000, S=16)
    assert [int(r["chain"]) for r in r0["recs"]] == list(range(15))
    for r, s in zip(r0["recs"], specs):                                  # == N independent single-rank runs
        assert (r["G"], r["seed"]) == (s["G"], s["seed"])
        assert r["lp_star"] == -1000.0 * s["G"] - s["seed"] and r["mean_dev"] == 2000.0 * s["G"] + s["seed"]
        assert r["G_final"] == s["G"] - (s["seed"] % 2)
    d = tmp_path / "Dev.csv"
    chains.write_dev_csv(str(d), r0["recs"])
    rows = open(d).read().strip().split("\n")
    assert rows[0] == "H,G,LP,Dev" and len(rows) == 16 and rows[1] == "2,2,-2000.000000,4000.000000"


def test_two_rank_gloo_failed_chain_is_requeued_and_never_strands_the_gather(tmp_path):
    """one chain fails on every attempt, another only once: both ranks still complete the all_gather, the flaky chain
    is re-run (2 attempts, good record), the broken one is tried twice and reported failed with NaN fit values"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29659", os.path.join(HERE, "_gloo_worker.py"), str(tmp_path), "flaky"]
    subprocess.run(cmd, check=True, env=env, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r0 = json.load(open(tmp_path / "flaky0.json"))
    r1 = json.load(open(tmp_path / "flaky1.json"))
    assert len(r0["recs"]) == 15 and [int(r["chain"]) for r in r0["recs"]] == list(range(15))
    calls = dict(r0["calls"]); calls.update(r1["calls"])
    assert calls["3_1"] == 2 and calls["5_2"] == 2                      # one re-queue each, no more
    for a, b in zip(r0["recs"], r1["recs"]):                            # identical on both ranks (NaN-aware)
        assert all((x == y) or (x != x and y != y) for x, y in zip(a.values(), b.values()))
    by = {(int(r["G"]), int(r["seed"])): r for r in r0["recs"]}
    bad, flaky = by[(3, 1)], by[(5, 2)]
    assert bad["failed"] == 1.0 and np.isnan(bad["lp_star"]) and np.isnan(bad["mean_dev"])
    assert flaky["failed"] == 0.0 and flaky["lp_star"] == -5002.0
    assert sum(r["failed"] for r in r0["recs"]) == 1.0
    d = tmp_path / "Dev.csv"
    chains.write_dev_csv(str(d), r0["recs"])
    assert len(open(d).read().strip().split("\n")) == 15              # header + 14 chains: the failed one is skipped


def test_group_units_and_two_rank_gloo_batched_units(tmp_path):
    """--batch: replicate chains of a G value are scheduled (LPT over units) and run together; a unit whose batched run
    fails falls back to its chains one by one; the gather is what the unbatched run gives"""
    specs = chains.sweep_specs(range(2, 7), 3, V=1000, S=16)
    assert chains.group_units(specs, 1) == [[i] for i in range(15)]
    assert chains.group_units(specs, 2) == [[0, 1], [2], [3, 4], [5], [6, 7], [8], [9, 10], [11], [12, 13], [14]]
    assert chains.group_units(specs, 8) == [[0, 1, 2], [3, 4, 5], [6, 7, 8], [9, 10, 11], [12, 13, 14]]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29667", os.path.join(HERE, "_gloo_worker.py"), str(tmp_path), "batch"]
    subprocess.run(cmd, check=True, env=env, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r0 = json.load(open(tmp_path / "batch0.json"))
    r1 = json.load(open(tmp_path / "batch1.json"))
    assert r0["recs"] == r1["recs"] and [int(r["chain"]) for r in r0["recs"]] == list(range(15))
    for r, s in zip(r0["recs"], specs):
        assert (r["G"], r["seed"], r["failed"]) == (s["G"], s["seed"], 0.0) and r["lp_star"] == -1000.0 * s["G"] - s["seed"]
    units = sorted(tuple(map(tuple, u)) for u in r0["units"] + r1["units"])
    assert units == sorted([((g, 0), (g, 1)) for g in range(2, 7)])       # pairs batched, the third replicate runs alone
    assert not {tuple(map(tuple, u)) for u in r0["units"]} & {tuple(map(tuple, u)) for u in r1["units"]}


def test_single_process_failure_handling():
    n = {"k": 0}

    def run(spec):
        n["k"] += 1
        if spec["seed"] == 1:
            raise ValueError("boom")
        return dict(G=spec["G"], seed=spec["seed"], G_final=spec["G"], lp_star=-1.0, mean_dev=2.0, iters=1)
    recs = chains.run_chains(chains.sweep_specs([2], 3, 10, 4), run, concurrency=2)
    assert [r["failed"] for r in recs] == [0.0, 1.0, 0.0] and n["k"] == 4


def test_eight_rank_gloo_config5_plan(tmp_path):
    """the 8-GPU plan of BASELINE config 5 (55 chains, V = 50k, S = 96) walked by 8 real ranks (gloo): every rank runs exactly
    the chains the LPT plan gives it -- chain by chain and as batched units of the five replicates of a G value -- and every
    rank ends with all 55 records; the plan's predicted imbalance is the one scripts/bench_config5.py checks against measured
    chain times (profiles/r03_config5.json)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
           "--master-addr", "127.0.0.1", "--master-port", "29671", os.path.join(HERE, "_gloo_worker.py"), str(tmp_path), "cfg5"]
    subprocess.run(cmd, check=True, env=env, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    specs = chains.sweep_specs(range(2, 13), 5, V=50000, S=96)
    costs = [s["cost"] for s in specs]
    plan = chains.lpt_assign(costs, 8)
    units = chains.group_units(specs, 5)
    assert units == [list(range(5 * k, 5 * k + 5)) for k in range(11)]
    uplan = chains.lpt_assign([sum(costs[i] for i in u) for u in units], 8)
    rs = [json.load(open(tmp_path / ("cfg5_%d.json" % r))) for r in range(8)]
    for r in range(8):
        assert rs[r]["one"] == plan[r]                                        # LPT order within the rank, too
        assert rs[r]["batched"] == [i for ui in uplan[r] for i in units[ui]]
        assert rs[r]["recs"] == rs[0]["recs"] and rs[r]["recs_b"] == rs[0]["recs"]
    assert [int(x["chain"]) for x in rs[0]["recs"]] == list(range(55))
    load = [sum(costs[i] for i in b) for b in plan]
    assert max(load) / (sum(load) / 8) < 1.05                                 # chain by chain: within 5 % of perfect balance
    uload = [sum(costs[i] for ui in b for i in units[ui]) for b in uplan]
    assert max(uload) / (sum(uload) / 8) < 1.45                               # 11 units on 8 GPUs: two units on three of them


def test_sys_exit_of_a_chain_is_contained_and_short_batches_are_rejected():
    """the CLI reports bad input through sys.exit (a BaseException): such a chain must become a failed record, not kill the
    rank before the gather; a batch_fn that returns fewer records than chains must not misalign them"""
    import sys as _sys
    calls = {"one": 0, "batch": 0}

    def run(spec):
        calls["one"] += 1
        if spec["seed"] == 0:
            _sys.exit("desman: no samples above the coverage cut")
        return dict(G=spec["G"], seed=spec["seed"], G_final=spec["G"], lp_star=-1.0, mean_dev=2.0, iters=1)
    recs = chains.run_chains(chains.sweep_specs([2], 3, 10, 4), run)
    assert [r["failed"] for r in recs] == [1.0, 0.0, 0.0] and calls["one"] == 4
    for conc in (1, 2):
        calls["one"] = 0
        recs = chains.run_chains(chains.sweep_specs([2], 3, 10, 4), run, concurrency=conc)
        assert [r["failed"] for r in recs] == [1.0, 0.0, 0.0] and calls["one"] == 4

    def short_batch(group):
        calls["batch"] += 1
        if group[0]["G"] == 3:
            _sys.exit("bad unit")
        return [dict(G=sp["G"], seed=sp["seed"], G_final=sp["G"], lp_star=-7.0, mean_dev=2.0, iters=1) for sp in group[:-1]]
    calls["one"] = 0
    recs = chains.run_chains(chains.sweep_specs([2, 3], 2, 10, 4), run, batch_fn=short_batch, batch=2)
    assert calls["batch"] == 2
    # both units fell back to one by one (short result / sys.exit): records come from run(), seed 0 fails twice
    assert [(int(r["G"]), int(r["seed"]), r["failed"], r["lp_star"] == -1.0) for r in recs] == \
        [(2, 0, 1.0, False), (2, 1, 0.0, True), (3, 0, 1.0, False), (3, 1, 0.0, True)]


# ---------------------------------------------------------------- row f4: genes sharded over ranks
def test_gene_partition_properties():
    from desman_amd.gene_shards import partition_genes
    rows = np.random.default_rng(0).integers(0, 40, size=500)
    for world in (1, 2, 3, 8):
        b = partition_genes(rows, world)
        assert b[0] == 0 and b[-1] == 500 and len(b) == world + 1 and (np.diff(b) >= 0).all()
        cost = rows + 4
        load = [cost[b[r]:b[r + 1]].sum() for r in range(world)]
        assert max(load) <= cost.sum() / world + cost.max()
    assert partition_genes([5, 5], 4).tolist()[-1] == 2                 # more ranks than genes: empty blocks allowed
    assert partition_genes(np.zeros(0), 2).tolist() == [0, 0, 0]


def test_two_rank_gloo_gene_gather(tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29657", os.path.join(HERE, "_gloo_gene_worker.py"), str(tmp_path)]
    subprocess.run(cmd, check=True, env=env, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r0 = json.load(open(tmp_path / "generank0.json"))
    r1 = json.load(open(tmp_path / "generank1.json"))
    assert r0 == r1                                                      # every rank holds the whole table
    rows = np.random.default_rng(3).integers(0, 30, size=41)
    assert r0["eta"] == list(range(41))                                  # genes in global order
    assert r0["tau"] == list(range(int(rows.sum())))                     # variant rows in global order
    assert r0["bounds"][0] == 0 and r0["bounds"][-1] == 41


def test_two_rank_gloo_work_queue_runs_every_chain_once_and_follows_the_faster_rank(tmp_path):
    """--schedule queue: the ranks draw chains from one counter (here the rendezvous store).  The slow rank ends up with fewer
    chains, every chain runs exactly once (the one that fails once is retried where it was drawn), both ranks gather all records"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29671", os.path.join(HERE, "_gloo_worker.py"), str(tmp_path), "queue"]
    subprocess.run(cmd, check=True, env=env, timeout=300, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    r0 = json.load(open(tmp_path / "queue0.json"))
    r1 = json.load(open(tmp_path / "queue1.json"))
    assert r0["recs"] == r1["recs"] and [int(r["chain"]) for r in r0["recs"]] == list(range(15))
    assert not any(r["failed"] for r in r0["recs"])
    first = lambda ran: list(dict.fromkeys(ran))                          # (the retried chain appears twice on its rank)
    assert sorted(first(r0["ran"]) + first(r1["ran"])) == list(range(15))
    assert len(first(r0["ran"])) > len(first(r1["ran"]))                  # the fast rank took more of the list
    specs = chains.sweep_specs(range(2, 7), 3, V=1000, S=16)
    for r, s in zip(r0["recs"], specs):                                  # the records of the static plan, whoever ran the chain
        assert (r["G"], r["seed"], r["lp_star"]) == (s["G"], s["seed"], -1000.0 * s["G"] - s["seed"])


def test_work_queue_file_counter_over_processes_and_threads(tmp_path):
    """the flock()ed counter file desman_amd.launch.spawn_ranks hands to its ranks: three processes x four threads draw 0..N-1
    exactly once each"""
    from desman_amd import launch
    prog = tmp_path / "draw.py"
    prog.write_text("import os, sys, threading\n"
                    "sys.path.insert(0, %r)\n"
                    "from desman_amd.chains import WorkQueue\n"
                    "q = WorkQueue(file_path=os.environ['DESMAN_SWEEP_QUEUE']); got = []\n"
                    "def w():\n"
                    "    while True:\n"
                    "        k = q.next()\n"
                    "        if k >= 600: return\n"
                    "        got.append(k)\n"
                    "th = [threading.Thread(target=w) for _ in range(4)]\n"
                    "[t.start() for t in th]; [t.join() for t in th]\n"
                    "open(os.path.join(sys.argv[1], 'got%%s' %% os.environ['RANK']), 'w').write(' '.join(map(str, got)))\n"
                    % os.path.dirname(HERE))
    assert launch.spawn_ranks(3, [sys.executable, str(prog), str(tmp_path)]) == 0
    got = [int(x) for r in range(3) for x in open(tmp_path / ("got%d" % r)).read().split()]
    assert sorted(got) == list(range(600))
    q = chains.WorkQueue()                                               # the in-process carrier
    assert [q.next() for _ in range(5)] == [0, 1, 2, 3, 4]


def test_work_queue_serves_a_second_call_and_a_dead_worker_is_not_silent(tmp_path):
    """(i) a second run_chains over the same queue gets its work (the counter belongs to the call: own 8 bytes of the
    file / own key of the store); (ii) a worker thread that dies in take() (flock / store error) no longer yields a short gather in
    silence: the chains it never ran come back as failed records and the error is raised"""
    specs = chains.sweep_specs(range(2, 5), 2, V=100, S=8)
    run = lambda sp: dict(G=sp["G"], seed=sp["seed"], G_final=sp["G"], lp_star=-1.0, mean_dev=2.0, iters=1)
    qfile = tmp_path / "q"
    qfile.write_bytes(b"\0" * 8)
    for q in (chains.WorkQueue(), chains.WorkQueue(file_path=str(qfile))):
        for _ in range(3):
            recs = chains.run_chains(specs, run, queue=q, concurrency=2)
            assert [int(r["chain"]) for r in recs] == list(range(len(specs))) and not any(r["failed"] for r in recs)

    class Broken(chains.WorkQueue):
        def next(self):
            k = super().next()
            if k == 2:
                raise OSError("store went away")
            return k

    with pytest.raises(RuntimeError, match="worker thread"):
        chains.run_chains(specs, run, queue=Broken(), concurrency=1)


def test_cost_model_matches_the_round6_measurements():
    """chain_cost against what bench.py measured at V = 50k, S = 96 (profiles/r06_chain_cost_components.json): every G within 5 %
    for the Gibbs iteration and 8 % for the NMF update; whole chains within 10 % where G fits the table (chains with too few or too
    many haplotypes cost more than any shape-only estimate says -- scripts/misfit_scan.py -- which is the reason for the work queue)"""
    prof = os.path.join(os.path.dirname(HERE), "profiles", "r06_chain_cost_components.json")
    d = json.load(open(prof))["per_G"]
    V, S = 50000, 96
    for g, row in d.items():
        g = int(g)
        assert abs(chains.chain_cost(V, S, g) / (1e3 * row["gibbs_ms_per_iter"]) - 1.0) < 0.05, g
        nmf = (chains.chain_cost(V, S, g, n_iter=0, nmf_updates=1) - chains.chain_cost(V, S, g, n_iter=0, nmf_updates=0))
        assert abs(nmf / row["nmft_us_per_update"] - 1.0) < 0.08, g
    # whole `desman -i 500` chains on the six-strain config-5 table (profiles/r06_chain_phases.txt, seconds): the chains that fit
    wall = {4: 0.87, 6: 1.02}
    for g in wall:
        assert abs(chains.chain_cost(V, S, g, n_iter=500) * 1e-6 / wall[g] - 1.0) < 0.10, g
    # ... and the config-3 line (profiles/r06_bench.json: 0.1004 ms per iteration, 23.0 us per NMF update)
    assert abs(chains.chain_cost(10000, 64, 8) / 100.4 - 1.0) < 0.06
    nmf3 = chains.chain_cost(10000, 64, 8, n_iter=0, nmf_updates=1) - chains.chain_cost(10000, 64, 8, n_iter=0, nmf_updates=0)
    assert abs(nmf3 / 23.0 - 1.0) < 0.08
