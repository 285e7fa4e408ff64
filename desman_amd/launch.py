"""One-command N-GPU launch: `python bench.py --gpus N` / `desman-sweep --gpus N` start their own N ranks.

The reference's fan-out is N background shell jobs (scripts/runDesman.sh:15-21).  Here a program that asks for N GPUs is
either already one of N ranks (started by `python -m torch.distributed.run --nproc-per-node N ...`: RANK / LOCAL_RANK /
WORLD_SIZE in the environment) or a plain process, which then replaces itself by that very launch.  In both cases the
world MUST be N: a mismatch is an error with a non-zero exit code, never a silent single-GPU run that prints a number.
"""
import os
import socket
import sys

EXIT_BAD_WORLD = 2


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus(use_torch=True):
    """number of GPUs this process could bind (0 without a GPU / driver).  Does not create a HIP context."""
    if not use_torch:
        try:
            from . import _lib
            return max(0, int(_lib.device_count()))
        except Exception:                                      # noqa: BLE001 -- library not built / no driver
            return 0
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:                                          # noqa: BLE001 -- no torch / no driver: no GPUs
        return 0


def _die(prog, msg):
    sys.stderr.write("%s: error: %s\n" % (prog, msg))
    sys.stderr.flush()
    raise SystemExit(EXIT_BAD_WORLD)


def ensure_world(gpus, argv, script=None, module=None, prog=None, n_visible=None, _exec=os.execve, launcher="torchrun"):
    """Make this process one of exactly `gpus` ranks, or exit non-zero saying why.

    gpus    -- what `--gpus` asked for (>= 1)
    argv    -- the program's own arguments (forwarded unchanged to every rank)
    script  -- path of the program (`bench.py`) or module= its module name (`desman_amd.chains`)
    Returns (rank, local_rank, world, launched_by_torchrun).  Does not return when it re-executes itself.

    * RANK in the environment (torch.distributed.run started us): WORLD_SIZE must equal `gpus`.
    * no RANK, gpus == 1: a plain single process, world of one, no process group.
    * no RANK, gpus > 1: needs `gpus` visible devices, then os.execve of
      `python -m torch.distributed.run --nnodes=1 --nproc-per-node gpus --master-addr 127.0.0.1 --master-port <free> <program> argv`
      (launcher="torchrun"), or -- launcher="spawn", no torch involved -- N children started by spawn_ranks() and this process
      exits with their status.
    """
    prog = prog or (os.path.basename(script) if script else module)
    if gpus < 1:
        _die(prog, "--gpus must be >= 1 (got %d)" % gpus)
    env = os.environ
    if "RANK" in env:
        world = int(env.get("WORLD_SIZE", "1"))
        if world != gpus:
            _die(prog, "--gpus %d but the launcher started a world of %d rank(s) (WORLD_SIZE=%s, RANK=%s): start it with "
                       "--nproc-per-node %d, or run `%s --gpus %d` as a plain process and it starts its own ranks"
                 % (gpus, world, env.get("WORLD_SIZE"), env.get("RANK"), gpus, prog, gpus))
        return int(env["RANK"]), int(env.get("LOCAL_RANK", "0")), world, True
    if gpus == 1:
        return 0, 0, 1, False
    n = visible_gpus(use_torch=(launcher == "torchrun")) if n_visible is None else n_visible
    if n < gpus:
        _die(prog, "--gpus %d asked for, %d GPU(s) visible on this node: refusing to run (a smaller run would print a number "
                   "that is not the %d-GPU number)" % (gpus, n, gpus))
    if launcher == "spawn":                                    # no torch anywhere: our own N children (spawn_ranks below)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        e = dict(env)
        e["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
        sys.stdout.flush()
        sys.stderr.flush()
        raise SystemExit(spawn_ranks(gpus, [sys.executable] + (["-m", module] if module else [os.path.abspath(script)]) + list(argv), env=e))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port())]
    cmd += (["-m", module] if module else [os.path.abspath(script)]) + list(argv)
    child_env = dict(env, MASTER_ADDR="127.0.0.1")
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL across processes needs it on this driver
    child_env.setdefault("OMP_NUM_THREADS", "1")
    if module:                                                 # `-m desman_amd.chains` must resolve wherever the ranks start
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        child_env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    sys.stdout.flush()
    sys.stderr.flush()
    _exec(sys.executable, cmd, child_env)
    raise AssertionError("unreachable: execve returned")       # pragma: no cover


def dist_backend():
    """the carrier of torch.distributed's collectives: DESMAN_DIST_BACKEND = "nccl" (default: RCCL over xGMI, one rank per GPU) or "gloo"
    (TCP between the ranks' hosts sides: the only carrier that lets two ranks share ONE device -- RCCL refuses a communicator with two
    ranks on a device -- which is what the two-rank rehearsal on a one-GPU box needs, tests/test_gpu_dist.py)"""
    b = os.environ.get("DESMAN_DIST_BACKEND", "nccl").strip().lower()
    if b not in ("nccl", "gloo"):
        _die("desman_amd.launch", "DESMAN_DIST_BACKEND=%r: nccl or gloo" % b)
    return b


def bind_device(local_rank, n_visible, prog="desman_amd"):
    """the device index a rank binds: its local rank, or -- only with DESMAN_DIST_SHARE_GPU=1 and the gloo carrier -- local rank modulo
    the number of devices (several ranks on one GPU: a rehearsal of the N > 1 control path, never a benchmark configuration)"""
    if local_rank < n_visible:
        return local_rank
    if os.environ.get("DESMAN_DIST_SHARE_GPU") == "1" and n_visible > 0 and dist_backend() == "gloo":
        return local_rank % n_visible
    _die(prog, "local rank %d but %d GPU(s) visible (several ranks on one device need DESMAN_DIST_BACKEND=gloo DESMAN_DIST_SHARE_GPU=1)"
         % (local_rank, n_visible))


def bound_device_record(rank, local_rank):
    """what a rank reports about the GPU it bound: index, name, PCI bus id / uuid where torch exposes them"""
    rec = dict(rank=int(rank), local_rank=int(local_rank), pid=os.getpid())
    try:
        import torch
        p = torch.cuda.get_device_properties(local_rank)
        rec.update(device_index=int(local_rank), name=str(p.name))
        for k in ("pci_bus_id", "pci_device_id", "uuid", "gcnArchName"):
            v = getattr(p, k, None)
            if v is not None:
                rec[k] = str(v)
    except Exception as e:                                     # noqa: BLE001
        rec["device_error"] = "%s: %s" % (type(e).__name__, e)
    return rec


def spawn_ranks(n, cmd, env=None, port=None):
    """start `cmd` (argv list) as n ranks of one node WITHOUT torch: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in
    each child's environment (what desman_amd/comm.py: Comm.from_env reads), stdout / stderr inherited.  Waits for all; when one rank
    fails the others are terminated (they would wait in a collective for ever).  Returns the first non-zero exit status, or 0."""
    import subprocess
    import tempfile
    import time
    port = port or free_port()
    procs = []
    # the counter of the chain scheduler's work queue (desman_amd/chains.py: WorkQueue): a fresh 8-byte file per launch, the
    # ranks take numbers from it under flock().  One node, one file system: the path's only topology (scripts/runDesman.sh:15-21).
    qfd, qpath = tempfile.mkstemp(prefix="desman_queue_")
    os.write(qfd, b"\0" * 8)
    os.close(qfd)
    for r in range(n):
        e = dict(os.environ if env is None else env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                 MASTER_PORT=str(port), DESMAN_SWEEP_QUEUE=qpath)
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        e.setdefault("OMP_NUM_THREADS", "1")
        procs.append(subprocess.Popen(list(cmd), env=e))
    rc = 0
    live = set(range(n))
    while live:
        for r in sorted(live):
            st = procs[r].poll()
            if st is None:
                continue
            live.discard(r)
            if st != 0 and rc == 0:
                rc = st
                for q in live:                                 # exact PIDs we started
                    procs[q].terminate()
        time.sleep(0.05)
    try:
        os.unlink(qpath)
    except OSError:
        pass
    return rc


def _main(argv):
    """python -m desman_amd.launch -n N program.py [args...]   (or: -n N -m package.module [args...])"""
    import argparse
    ap = argparse.ArgumentParser(prog="python -m desman_amd.launch", description="start N ranks of a program on this node, no torch")
    ap.add_argument("-n", "--nproc", type=int, required=True)
    ap.add_argument("-m", dest="module", default=None)
    ap.add_argument("rest", nargs=argparse.REMAINDER)
    a = ap.parse_args(argv)
    cmd = [sys.executable] + (["-m", a.module] if a.module else []) + a.rest
    raise SystemExit(spawn_ranks(a.nproc, cmd))


if __name__ == "__main__":
    _main(sys.argv[1:])
