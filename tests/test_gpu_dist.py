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
