#!/usr/bin/env python3
"""What sharding ONE chain by positions costs and what it can bring (desman_amd/vshard.py): run under torch.distributed.run with N ranks
(N = 1 here: no multi-GPU box was available to the builder).  Prints ms per iteration of the unsharded chain, of the sharded chain on N
ranks, and -- from the unsharded chain's per-kernel times -- the iteration time an N-GPU run is expected to take: the position-parallel
kernels (stage 1 of the mu/E pass, tau sweep) divided by N, the replicated ones (stage 2 / Dirichlet) and the measured exchange added.
usage: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P scripts/bench_vshard.py [V S G iters]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from desman_amd import _lib, vshard  # noqa: E402
from desman_amd.synth import synth_counts  # noqa: E402
from oracle import cbind  # noqa: E402  (idx -> one-hot helper only)

a = sys.argv[1:]
V, S, G = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (50000, 96, 12)
n_iter = int(a[3]) if len(a) > 3 else 100
local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
world, rank = dist.get_world_size(), dist.get_rank()
counts, tau_true, gamma_true = synth_counts(V, S, G, seed=1234)
tau, gamma, eta = cbind.idx_to_onehot(tau_true), np.ascontiguousarray(gamma_true), 0.96 * np.eye(4) + 0.01
b = vshard.shard_bounds(V, world)
ch = vshard.ShardedChain(counts[b[rank]:b[rank + 1]], b[rank], V, G, 1, device=local, ctr_seed=5)
ch.set_state(tau[b[rank]:b[rank + 1]], gamma, eta)
ex = vshard.TorchExchange(dist, torch.device("cuda", local))
ch.update(10, ex)
dist.barrier(); torch.cuda.synchronize()
t0 = time.perf_counter(); ch.update(n_iter, ex); dt_sh = time.perf_counter() - t0
out = None
if rank == 0:
    c = _lib.Context(local); c.set_counts(counts); c.set_state(tau, gamma, eta); c.seed(1, ctr_seed=5); c.set_tau_rng(_lib.RNG_PHILOX)
    c.force_stats_spec(_lib.STATS_AGG); c.gibbs_update(10)
    t0 = time.perf_counter(); c.gibbs_update(n_iter); dt_un = time.perf_counter() - t0
    c.set_timing(True); c.gibbs_update(20); tm = c.get_timing(); c.set_timing(False)
    k = {n: 1e3 * ms / max(cnt, 1) for n, (ms, cnt) in tm.items() if cnt}
    par = k.get("stats", 0) + k.get("stats_big", 0) + k.get("tau", 0)
    rep = k.get("dirichlet", 0) + k.get("stats2", 0)
    out = dict(V=V, S=S, G=G, ranks=world, iters=n_iter, ms_per_iter_unsharded=1e3 * dt_un / n_iter, ms_per_iter_sharded=1e3 * dt_sh / n_iter,
               exchange_overhead_us_at_this_world=1e6 * (dt_sh - dt_un) / n_iter if world == 1 else None,
               kernels_us=k, table_bytes=int((1 << G) * S * 4),
               expected_ms_per_iter={n: (par / n + rep) / 1e3 + max(0.0, (dt_sh - dt_un) / n_iter * 1e3) if world == 1 else None for n in (2, 4, 8)},
               note="expected = (stage 1 + tau sweep) / N + stage 2 + Dirichlet + the exchange overhead measured with one rank (host sync + two "
                    "RCCL calls; a real N-rank all-reduce of the table adds its xGMI time, ~tens of us for MBs); no N > 1 run has been made")
    print(json.dumps(out))
dist.destroy_process_group()
