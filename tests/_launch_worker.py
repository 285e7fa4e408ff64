"""Worker of tests/test_launch_cpu.py: a program with a `--gpus N` flag that goes through desman_amd.launch.ensure_world exactly as
bench.py / desman-sweep do; the ranks talk over gloo (no GPU here) and rank 0 prints one JSON line."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from desman_amd import launch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--gpus", type=int, default=1)
ap.add_argument("--pretend-visible", type=int, default=None)
ap.add_argument("--tag", default="")
args = ap.parse_args()
rank, local, world, under = launch.ensure_world(args.gpus, sys.argv[1:], script=__file__, n_visible=args.pretend_visible)
ranks = [dict(rank=rank, local_rank=local, pid=os.getpid())]
if under:
    import torch.distributed as dist
    dist.init_process_group("gloo")
    assert dist.get_world_size() == args.gpus
    ranks = [None] * world
    dist.all_gather_object(ranks, dict(rank=rank, local_rank=local, pid=os.getpid()))
    dist.destroy_process_group()
if rank == 0:
    print(json.dumps(dict(n_gpus=world, under_launcher=under, ranks=ranks, tag=args.tag)))
