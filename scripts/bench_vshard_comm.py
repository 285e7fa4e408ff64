#!/usr/bin/env python3
"""One chain sharded by positions with the exchange done INSIDE the library (dsm_ctx_gibbs_update_sharded_comm: grouped RCCL
all-reduces enqueued on the chain's stream; desman_amd/comm.py, no torch): per-iteration overhead against the unsharded chain, and the
N-GPU iteration time expected from the per-kernel times.  Plain process = a world of one; N ranks: `python -m desman_amd.launch -n N
scripts/bench_vshard_comm.py ...` (or torch.distributed.run).  usage: bench_vshard_comm.py [V S G iters]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from desman_amd import _lib, vshard  # noqa: E402
from desman_amd.comm import Comm  # noqa: E402
from desman_amd.synth import synth_counts  # noqa: E402

a = sys.argv[1:]
V, S, G = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (50000, 96, 12)
n_iter = int(a[3]) if len(a) > 3 else 100
comm = Comm.from_env()
rank, world, local = comm.rank, comm.world, comm.device
counts, tau_true, gamma_true = synth_counts(V, S, G, seed=1234)
tau = np.zeros((V, G, 4), dtype=np.int64)
np.put_along_axis(tau, tau_true[..., None].astype(np.int64), 1, axis=2)
gamma, eta = np.ascontiguousarray(gamma_true), 0.96 * np.eye(4) + 0.01
b = vshard.shard_bounds(V, world)
ch = vshard.ShardedChain(counts[b[rank]:b[rank + 1]], b[rank], V, G, 1, device=local, ctr_seed=5)
ch.set_state(tau[b[rank]:b[rank + 1]], gamma, eta)
ch.update(20, comm)
times = []
for _ in range(5):
    comm.barrier()
    t0 = time.perf_counter(); ch.update(n_iter, comm); times.append(time.perf_counter() - t0)
dt_sh = float(comm.allreduce(np.array([np.median(times)]), "max")[0])
if rank == 0:
    c = _lib.Context(local); c.set_counts(counts); c.set_state(tau, gamma, eta); c.seed(1, ctr_seed=5); c.set_tau_rng(_lib.RNG_PHILOX)
    c.force_stats_spec(_lib.STATS_AGG); c.gibbs_update(20)
    tu = []
    for _ in range(5):
        t0 = time.perf_counter(); c.gibbs_update(n_iter); tu.append(time.perf_counter() - t0)
    dt_un = float(np.median(tu))
    c.set_timing(True); c.gibbs_update(20); tm = c.get_timing(); c.set_timing(False)
    k = {n: 1e3 * ms / max(cnt, 1) for n, (ms, cnt) in tm.items() if cnt}
    par = k.get("stats", 0) + k.get("stats_big", 0) + k.get("tau", 0)
    rep = k.get("dirichlet", 0) + k.get("stats2", 0)
    ovh = 1e6 * (dt_sh - dt_un) / n_iter if world == 1 else None
    print(json.dumps(dict(V=V, S=S, G=G, ranks=world, iters=n_iter, ms_per_iter_unsharded=1e3 * dt_un / n_iter, ms_per_iter_sharded=1e3 * dt_sh / n_iter,
                          exchange_overhead_us_world_of_one=ovh, kernels_us={n: round(v, 1) for n, v in k.items()}, table_bytes=int((1 << G) * S * 4),
                          expected_ms_per_iter={n: (par / n + rep + max(ovh, 0.0)) / 1e3 for n in (2, 4, 8)} if world == 1 else None,
                          exchange="dsm_ctx_gibbs_update_sharded_comm: ncclGroupStart / 2 x ncclAllReduce / ncclGroupEnd on the chain's stream, no host sync",
                          note="expected = (stage 1 + tau sweep) / N + stage 2 + Dirichlet + the overhead measured with one rank; a real N-rank "
                               "all-reduce adds its xGMI time; NO N > 1 RUN HAS BEEN MADE")))
comm.close()
