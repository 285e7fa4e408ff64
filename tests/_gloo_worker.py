"""worker for tests/test_chains_gloo.py: world_size-2 run of the chain scheduler on CPU."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist  # noqa: E402

from desman_amd import chains  # noqa: E402


def fake_run(spec):
    # deterministic stand-in for a GPU chain: the scheduler/gather is what is under test
    g, s = spec["G"], spec["seed"]
    return dict(G=g, seed=s, G_final=g - (s % 2), lp_star=-1000.0 * g - s, mean_dev=2000.0 * g + s, iters=10,
                wall_s=0.001 * g)


if __name__ == "__main__":
    dist.init_process_group("gloo")
    specs = chains.sweep_specs(range(2, 7), 3, V=1000, S=16)
    recs = chains.run_chains(specs, fake_run, dist)
    bins = chains.lpt_assign([s["cost"] for s in specs], dist.get_world_size())
    out = dict(rank=dist.get_rank(), recs=recs, mine=bins[dist.get_rank()])
    with open(os.path.join(sys.argv[1], "rank%d.json" % dist.get_rank()), "w") as f:
        json.dump(out, f)
    dist.destroy_process_group()
