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


def visible_gpus():
    """number of GPUs this process could bind (0 without a GPU / driver).  Does not create a HIP context."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:                                          # noqa: BLE001 -- no torch / no driver: no GPUs
        return 0


def _die(prog, msg):
    sys.stderr.write("%s: error: %s\n" % (prog, msg))
    sys.stderr.flush()
    raise SystemExit(EXIT_BAD_WORLD)


def ensure_world(gpus, argv, script=None, module=None, prog=None, n_visible=None, _exec=os.execve):
    """Make this process one of exactly `gpus` ranks, or exit non-zero saying why.

    gpus    -- what `--gpus` asked for (>= 1)
    argv    -- the program's own arguments (forwarded unchanged to every rank)
    script  -- path of the program (`bench.py`) or module= its module name (`desman_amd.chains`)
    Returns (rank, local_rank, world, launched_by_torchrun).  Does not return when it re-executes itself.

    * RANK in the environment (torch.distributed.run started us): WORLD_SIZE must equal `gpus`.
    * no RANK, gpus == 1: a plain single process, world of one, no process group.
    * no RANK, gpus > 1: needs `gpus` visible devices, then os.execve of
      `python -m torch.distributed.run --nnodes=1 --nproc-per-node gpus --master-addr 127.0.0.1 --master-port <free> <program> argv`.
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
    n = visible_gpus() if n_visible is None else n_visible
    if n < gpus:
        _die(prog, "--gpus %d asked for, %d GPU(s) visible on this node: refusing to run (a smaller run would print a number "
                   "that is not the %d-GPU number)" % (gpus, n, gpus))
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
