"""ONE Gibbs chain over several GPUs, sharded by positions (last row).

The reference has no such mode -- its only parallel axis is independent chains (desman_amd/chains.py).  It is for the case
the chain scheduler cannot help: fewer chains than GPUs on a large table (V >~ 50k), where an iteration is long enough to pay
one exchange.  Rank r holds a contiguous slice of the positions (count tensor and tau); gamma / eta are replicated.  Per
iteration every rank runs stage 1 of the auxiliary-count pass on its slice, the ranks all-reduce the subset table (uint32
[2^G][S]) and an 18-double vector (RCCL over xGMI: `torch.distributed`, backend "nccl"), then stage 2, the gamma / eta draws
(replicated: same inputs, same counter-based streams -> same bits on every rank) and the tau sweep of the rank's slice.
Every draw is keyed by GLOBAL indices, so the chain does not depend on the number of ranks: it is the unsharded chain UNDER THE
AGGREGATED mu/E PASS, SPEC 2 (dsm_ctx_force_stats_spec(ctx, 2) / DESMAN_HIP_STATS_SPEC=2), bit for bit (tests/test_gpu_vshard.py:
two / three shards on one GPU against one context; tests/test_gpu_fullsize.py at 50k x 96).  By the shape rule an unsharded chain
may run another specification of the same law -- the per-read pass on small or shallow tables, the word-pooled pass (spec 4) on large
tables with up to eight haplotypes, whose representatives would be per shard -- and then draws other variates than its sharded form.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... your_script.py
        chain = ShardedChain(counts_slice, v_offset, v_total, G, seed, device=local_rank)
        chain.set_state(tau_slice, gamma, eta)
        chain.update(n_iter, Comm.from_env())                 # the library's own RCCL exchange (desman_amd/comm.py), no torch
        chain.update(n_iter, TorchExchange(dist, device))     # or through torch.distributed (host-synchronised per iteration)
"""
import ctypes as C

import numpy as np

from . import _lib


def shard_bounds(v_total, world):
    """[b_0 = 0, ..., b_world = v_total]: contiguous slices of (nearly) equal size"""
    return [(v_total * r) // world for r in range(world + 1)]


class _DevView:
    """a device buffer as an object `torch.as_tensor` understands (__cuda_array_interface__, version 2)"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class TorchExchange:
    """the exchange of dsm_ctx_gibbs_update_sharded as two in-place all-reduces on the device buffers (backend "nccl" = RCCL).
    uint32 sums are done as int32 (two's complement: the same bits)."""

    def __init__(self, dist, device):
        import torch
        self.torch, self.dist, self.device = torch, dist, device
        self.calls = 0
        self._n_tab_checked = None

    def __call__(self, tab_ptr, n_tab, vec_ptr, n_vec):
        torch = self.torch
        if n_tab and n_tab != self._n_tab_checked:
            # every rank must hand over a table of the same length (the library derives its layout from v_total, S, G): a mismatch
            # would hang or corrupt the all-reduce, so it is checked once per length, loudly
            t = torch.tensor([n_tab, -n_tab], device=self.device, dtype=torch.int64)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            if int(t[0]) != n_tab or int(-t[1]) != n_tab:
                raise _lib.DesmanHipError("sharded chain: subset tables of different length on different ranks (%d here, %d..%d over "
                                          "the ranks)" % (n_tab, int(-t[1]), int(t[0])))
            self._n_tab_checked = n_tab
        if n_tab:
            t = torch.as_tensor(_DevView(tab_ptr, n_tab, "<i4"), device=self.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        v = torch.as_tensor(_DevView(vec_ptr, n_vec, "<f8"), device=self.device)
        self.dist.all_reduce(v, op=self.dist.ReduceOp.SUM)
        torch.cuda.synchronize(self.device)
        self.calls += 1


class HostExchange:
    """the same reduction through host memory for shards that live in ONE process (one thread per shard): tests and single-GPU
    dry runs.  Every shard's callback deposits its buffers, the last one to arrive sums them, all pick the sums up."""

    def __init__(self, n_shards, device=0, timeout=600.0):
        import threading
        self.n, self.device = n_shards, device
        # a shard thread that fails before reaching the barrier must not hang its peers: the waits time out (BrokenBarrierError)
        self.bar = threading.Barrier(n_shards, timeout=timeout)
        self.lock = threading.Lock()
        self.tabs, self.vecs = {}, {}
        self.sum_tab = self.sum_vec = None

    def for_shard(self, k):
        lib = _lib.load()

        def exchange(tab_ptr, n_tab, vec_ptr, n_vec):
            tab = np.empty(n_tab, dtype=np.uint32)
            vec = np.empty(n_vec, dtype=np.float64)
            if n_tab:
                _lib.check(lib.dsm_device_read(self.device, tab_ptr, tab.ctypes.data, tab.nbytes))
            _lib.check(lib.dsm_device_read(self.device, vec_ptr, vec.ctypes.data, vec.nbytes))
            with self.lock:
                self.tabs[k], self.vecs[k] = tab, vec
            if self.bar.wait() == 0:
                if len({t.size for t in self.tabs.values()}) != 1:
                    self.bar.abort()
                    raise _lib.DesmanHipError("sharded chain: subset tables of different length on different shards: %s"
                                              % sorted(t.size for t in self.tabs.values()))                       # one thread reduces, in shard order (fixed order: reproducible sums)
                self.sum_tab = sum((self.tabs[j] for j in range(self.n)), np.zeros(n_tab, dtype=np.uint32)) if n_tab else None
                acc = np.zeros(n_vec)
                for j in range(self.n):
                    acc = acc + self.vecs[j]
                self.sum_vec = acc
            self.bar.wait()
            if n_tab:
                st = np.ascontiguousarray(self.sum_tab)
                _lib.check(lib.dsm_device_write(self.device, tab_ptr, st.ctypes.data, st.nbytes))
            _lib.check(lib.dsm_device_write(self.device, vec_ptr, self.sum_vec.ctypes.data, self.sum_vec.nbytes))
            self.bar.wait()
        return exchange


class ShardedChain:
    """this rank's shard of the chain: a device context holding positions v_offset .. v_offset + V of v_total"""

    def __init__(self, counts_slice, v_offset, v_total, G, seed, device=0, ctr_seed=None):
        counts_slice = np.ascontiguousarray(counts_slice, dtype=np.int64)
        self.V, self.S, self.G = counts_slice.shape[0], counts_slice.shape[1], int(G)
        self.v_offset, self.v_total = int(v_offset), int(v_total)
        self.ctx = _lib.Context(device)
        self.ctx.set_counts(counts_slice)
        self._seed = (seed, ctr_seed)

    def set_state(self, tau_slice, gamma, eta):
        self.ctx.set_state(np.ascontiguousarray(tau_slice, dtype=np.int64), np.ascontiguousarray(gamma, dtype=np.float64),
                           np.ascontiguousarray(eta, dtype=np.float64))
        self.ctx.seed(self._seed[0], ctr_seed=self._seed[1])            # the SAME counter-stream key on every shard
        self.ctx.set_tau_rng(_lib.RNG_PHILOX)

    def update(self, n_iter, exchange):
        """exchange: a desman_amd.comm.Comm (the library's own RCCL all-reduces, enqueued on the chain's stream: no host
        synchronisation per iteration) or a callable (TorchExchange, HostExchange.for_shard(k): called with the stream drained)"""
        from .comm import Comm
        if isinstance(exchange, Comm):
            self.ctx.gibbs_update_sharded_comm(n_iter, self.v_offset, self.v_total, exchange)
        else:
            self.ctx.gibbs_update_sharded(n_iter, self.v_offset, self.v_total, exchange)

    def trace(self):
        return self.ctx.get_trace()

    def state(self):
        return self.ctx.get_state()

    def star(self):
        return self.ctx.get_star()

    def close(self):
        self.ctx.close()
