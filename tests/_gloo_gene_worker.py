"""worker for tests/test_chains_gloo.py: world_size-2 run of the gene sharding (partition + gather) on CPU."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

from desman_amd.gene_shards import gather_blocks, partition_genes  # noqa: E402

if __name__ == "__main__":
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rows = np.random.default_rng(3).integers(0, 30, size=41)
    b = partition_genes(rows, world)
    lo, hi = int(b[rank]), int(b[rank + 1])
    # stand-in for the per-gene / per-row results of this rank's sampler: functions of the GLOBAL indices
    r0 = int(rows[:lo].sum())
    local = {"eta_star": np.arange(lo, hi)[:, None] * np.ones((1, 3)),
             "tau_star": (r0 + np.arange(int(rows[lo:hi].sum())))[:, None, None] * np.ones((1, 3, 4), dtype=np.int64)}
    got = gather_blocks(local, dist)
    with open(os.path.join(sys.argv[1], "generank%d.json" % rank), "w") as f:
        json.dump(dict(bounds=b.tolist(), eta=got["eta_star"][:, 0].tolist(), tau=got["tau_star"][:, 0, 0].tolist()), f)
    dist.destroy_process_group()
