"""First-contact insurance for the N > 1 path, on ONE GPU: `torch.distributed.run` with a world of one rank walks exactly the
code the driver's 2/4/8-GPU runs take -- RCCL communicator set-up (backend "nccl"), barrier, MAX all-reduce of the step time,
all_gather of the fit records, Dev.csv writer -- so that what is left for the 8-GPU node to show is scaling, not whether the
path runs.  (No 1 -> 8 curve has been measured: the round-end SCALE run needs an 8-GPU node.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pandas as p
import pytest

from desman_amd.synth import synth_counts

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(args, port, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


def test_bench_walks_the_rccl_path_with_a_world_of_one():
    argv = ["bench.py", "--gpus", "1", "--steps", "6", "--warmup", "2", "--V", "2000", "--S", "16", "--G", "4", "--no-cpu-baseline",
            "--no-nmft", "--no-pmc", "--repeats", "3"]
    out = _torchrun(argv, 29731)
    line = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 6 and line["scaling"] == "weak"
    assert line["launch"].startswith("torch.distributed.run, 1 rank") and len(line["ranks"]) == 1
    r0 = line["ranks"][0]
    assert r0["rank"] == 0 and r0.get("device_index") == 0 and r0.get("name") and r0["ms_per_step"] > 0, r0
    rp = line["ms_per_step_repeats"]
    assert rp["n"] == 3 and rp["min"] <= rp["median"] <= rp["max"] and rp["median"] == line["ms_per_step"]
    # the same command as a plain process: the same keys (plus the single-GPU extras), no process group
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable] + argv + ["--batch", "0"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    plain = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert set(plain) == set(line) and plain["n_gpus"] == 1 and plain["launch"].startswith("single process")
    # and more GPUs than this box has: exit status 2, a message, no number (one GPU here)
    r = subprocess.run([sys.executable] + argv[:1] + ["--gpus", "2"] + argv[3:], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and "refusing to run" in r.stderr and "n_gpus" not in r.stdout
    out8 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                           "--master-port", "29735"] + argv[:1] + ["--gpus", "8"] + argv[3:], env=dict(env, MASTER_ADDR="127.0.0.1"),
                          cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out8.returncode != 0 and "world of 1 rank" in out8.stderr and "n_gpus" not in out8.stdout
    assert line["value"] > 0 and line["ms_per_step"] > 0
    assert len(line["fit_records"]) == 1 and line["fit_records"][0]["seed"] == 0          # the all_gather'ed record of rank 0
    assert np.isfinite(line["fit_records"][0]["lp_star"]) and "batch" not in line        # N-GPU runs time the one chain only
    assert line["roofline"]["tau_launch"]["resident_workgroups"] > 0


def test_desman_sweep_walks_the_rccl_gather_with_a_world_of_one(tmp_path):
    V, S = 300, 8
    counts, _, _ = synth_counts(V, S, 3, seed=21)
    cols = ["Position"] + ["S%d-%s" % (s, b) for s in range(S) for b in "ACGT"]
    data = np.concatenate([np.arange(V)[:, None] * 7 + 3, counts.reshape(V, S * 4)], axis=1)
    df = p.DataFrame(data, index=["contig%d" % (v // 50) for v in range(V)], columns=cols)
    df.index.name = "Contig"
    freq = str(tmp_path / "syn.freq")
    df.to_csv(freq)
    stub = str(tmp_path / "sw")
    out = _torchrun(["-m", "desman_amd.chains", freq, "--gmin", "2", "--gmax", "3", "--reps", "2", "-i", "15", "-o", stub, "-c", "1",
                     "--comm", "torch"], 29733)
    recs = json.loads([ln for ln in out.splitlines() if ln.startswith("[{")][-1])
    assert [(int(r["G"]), int(r["seed"]), r["failed"]) for r in recs] == [(2, 0, 0.0), (2, 1, 0.0), (3, 0, 0.0), (3, 1, 0.0)]
    rows = open(stub + "_Dev.csv").read().strip().split("\n")
    assert rows[0] == "H,G,LP,Dev" and len(rows) == 5
    # the default: the gather through the library's own RCCL communicator (dsm_comm_*), ranks started without torch
    # (python -m desman_amd.launch -n 1 = what `desman-sweep --gpus N` does for itself); no torch in those processes
    stub2 = str(tmp_path / "sw2")
    env2 = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r2 = subprocess.run([sys.executable, "-m", "desman_amd.launch", "-n", "1", "-m", "desman_amd.chains", freq, "--gpus", "1", "--gmin", "2",
                         "--gmax", "3", "--reps", "2", "-i", "15", "-o", stub2, "-c", "1"], env=env2, cwd=ROOT, capture_output=True, text=True,
                        timeout=600)
    assert r2.returncode == 0, r2.stderr[-3000:]
    assert open(stub + "_Dev.csv").read() == open(stub2 + "_Dev.csv").read()
    r3 = subprocess.run([sys.executable, "-m", "desman_amd.chains", freq, "--gpus", "2", "-o", stub2], env=env2, cwd=ROOT, capture_output=True,
                        text=True, timeout=600)
    assert r3.returncode == 2 and "refusing to run" in r3.stderr          # one GPU here: no silent single-GPU sweep
    # the same sweep without torch.distributed: the gather changes nothing (in a child process as well: the sweep driver
    # imports torch, whose bundled HIP runtime this test process should not load next to the library's)
    stub1 = str(tmp_path / "sw1")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    r = subprocess.run([sys.executable, "-m", "desman_amd.chains", freq, "--gmin", "2", "--gmax", "3", "--reps", "2", "-i", "15", "-o", stub1,
                        "-c", "1"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert open(stub + "_Dev.csv").read() == open(stub1 + "_Dev.csv").read()


def _torchrun2(args, port, timeout=900):
    """two ranks that share device 0: torch.distributed.run --nproc-per-node 2 with the gloo carrier"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", DESMAN_DIST_BACKEND="gloo", DESMAN_DIST_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + args
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout


def test_bench_two_ranks_share_one_gpu():
    """The first N > 1 run of bench.py must not be the driver's 8-GPU run: two ranks, both on device 0, walk everything in the N > 1 path
    that does not depend on RCCL -- the world asserted against --gpus, the process group, the barrier around the timed call, the MAX
    all-reduce of the step time, all_gather_object of the per-rank records, the gather of the fit records, ONE JSON line from rank 0.
    RCCL itself refuses a communicator with two ranks on one device (ncclInvalidUsage: duplicate GPU), so the carrier here is gloo
    (DESMAN_DIST_BACKEND=gloo, DESMAN_DIST_SHARE_GPU=1: desman_amd/launch.py); the RCCL set-up keeps its world-of-one test above.
    A rehearsal of control flow, never a number: the line says so in `launch`.  Reference fan-out: scripts/runDesman.sh:15-21."""
    argv = ["bench.py", "--gpus", "2", "--steps", "5", "--warmup", "2", "--V", "2000", "--S", "16", "--G", "4", "--no-cpu-baseline",
            "--no-nmft", "--no-pmc", "--repeats", "3"]
    out = _torchrun2(argv, 29741)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                                        # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 5 and line["scaling"] == "weak"
    assert "2 rank(s)" in line["launch"] and "gloo" in line["launch"]
    assert [r["rank"] for r in line["ranks"]] == [0, 1] and [r["local_rank"] for r in line["ranks"]] == [0, 1]
    assert line["ranks"][0]["pid"] != line["ranks"][1]["pid"]                     # two processes
    assert all(r.get("device_index") == 0 and r["ms_per_step"] > 0 for r in line["ranks"])      # ... on the one device
    assert line["ms_per_step"] >= max(r["ms_per_step"] for r in line["ranks"]) * 0.5           # the job's time is the MAX over ranks' calls
    fits = line["fit_records"]
    assert len(fits) == 2 and sorted(int(f["seed"]) for f in fits) == [0, 1]      # every rank's chain in the gathered records
    assert all(np.isfinite(f["lp_star"]) for f in fits)
    assert fits[0]["lp_star"] != fits[1]["lp_star"]                               # two chains (seeds 0 and 1), not one twice
    # the value is the aggregate over both ranks' chains
    assert line["value"] == pytest.approx(2 * 2000 * 16 / (line["ms_per_step"] * 1e-3), rel=1e-6)
    # without the switch the same launch refuses (local rank 1, one device): exit status != 0 and no line
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DESMAN_DIST_BACKEND", "DESMAN_DIST_SHARE_GPU"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29743"] + argv, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "n_gpus" not in r.stdout and "GPU(s) visible" in r.stderr


def test_desman_sweep_two_ranks_share_one_gpu(tmp_path):
    """desman-sweep with two ranks x two worker threads on one GPU (gloo carrier): the work queue shared by the ranks (a counter in the
    rendezvous store), every chain run exactly once by some thread of some rank, ONE gather, Dev.csv and the chains' files equal to the
    one-rank run's (a chain's result does not depend on where it ran).  Reference: scripts/runDesman.sh:15-21, complete_example/
    README.md:626-627 (the Dev.csv the sweep ends in)."""
    V, S = 300, 8
    counts, _, _ = synth_counts(V, S, 3, seed=21)
    cols = ["Position"] + ["S%d-%s" % (s, b) for s in range(S) for b in "ACGT"]
    data = np.concatenate([np.arange(V)[:, None] * 7 + 3, counts.reshape(V, S * 4)], axis=1)
    df = p.DataFrame(data, index=["contig%d" % (v // 50) for v in range(V)], columns=cols)
    df.index.name = "Contig"
    freq = str(tmp_path / "syn.freq")
    df.to_csv(freq)
    common = [freq, "--gmin", "2", "--gmax", "4", "--reps", "3", "-i", "12", "--comm", "torch"]
    stub2 = str(tmp_path / "two")
    out = _torchrun2(["-m", "desman_amd.chains"] + common + ["-o", stub2, "-c", "2", "--gpus", "2"], 29745)
    recs = json.loads([ln for ln in out.splitlines() if ln.startswith("[{")][-1])
    assert [(int(r["G"]), int(r["seed"]), r["failed"]) for r in recs] == [(g, s, 0.0) for g in (2, 3, 4) for s in range(3)]     # every chain once
    stub1 = str(tmp_path / "one")
    _torchrun(["-m", "desman_amd.chains"] + common + ["-o", stub1, "-c", "1"], 29747)
    dev2, dev1 = open(stub2 + "_Dev.csv").read(), open(stub1 + "_Dev.csv").read()
    assert dev2.splitlines()[0] == "H,G,LP,Dev" and len(dev2.strip().splitlines()) == 10
    assert dev2 == dev1
    for g in (2, 3, 4):
        for s in range(3):
            for name in ("fit.txt", "Gamma_star.csv", "Eta_star.csv", "Filtered_Tau_star.csv"):
                a = open(os.path.join("%s_%d_%d" % (stub2, g, s), name)).read()
                b = open(os.path.join("%s_%d_%d" % (stub1, g, s), name)).read()
                assert a == b, (g, s, name)
