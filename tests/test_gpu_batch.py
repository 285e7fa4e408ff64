"""Batched Gibbs loop (include/desman_hip.h: dsm_batch_gibbs_update): K chains of one shape share every launch of the
iteration.  Each chain of the batch must end exactly where dsm_ctx_gibbs_update leaves it (aggregated mu/E pass)."""
import numpy as np
import pytest

from desman_amd import _lib
from desman_amd.synth import synth_counts, random_state

pytestmark = pytest.mark.gpu


def _chain(counts, state, seed, ctr_seed):
    c = _lib.Context(0)
    c.set_counts(counts)
    c.set_state(*state)
    c.set_tau_rng(_lib.RNG_MT19937)
    c.seed(seed, ctr_seed=ctr_seed)
    return c


def _snapshot(c):
    tr = c.get_trace()
    star = c.get_star()
    out = dict(state=c.get_state(), mt=c.get_mt_state(), lp_star=star["lp"])
    out.update({k: tr[k] for k in ("ll", "lp", "nchange", "gamma", "eta")})
    out["tau_last"] = c.get_tau_at(len(tr["ll"]) - 1)
    return out


def _same(a, b):
    for k in a:
        if k == "state":
            assert all(np.array_equal(x, y) for x, y in zip(a[k], b[k])), k
        elif k == "lp_star":
            assert a[k] == b[k]
        else:
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


# shapes: fused stage 2 (G < 10) with lane groups of 16 / 32 / 64, the stand-alone stage-2 launch (G >= 10), several
# MT19937 chunks (n_iter > 1 + 2 + 3), a second call on the same contexts (stream positions, buffers kept).
# spec: 0 = nothing forced -- every chain, alone or in the batch, runs what the shape rule gives (round 5: a chain's draws do not depend
# on how it is run; these small tables take the per-read pass, which a batch launches chain by chain); 2 / 4 = every context asks for it
@pytest.mark.parametrize("spec", [0, 2, 4])
@pytest.mark.parametrize("V,S,G,K,n_iter", [(300, 16, 5, 3, 12), (200, 64, 8, 5, 9), (150, 40, 3, 8, 7), (90, 20, 11, 2, 5), (64, 100, 4, 4, 6),
                                             (900, 10, 2, 3, 5)])      # the last: subset table in 8 copies
def test_batch_equals_chains_run_one_by_one(V, S, G, K, n_iter, spec):
    counts, _, _ = synth_counts(V, S, G, seed=500 + V)
    states = [random_state(V, S, G, seed=600 + k) for k in range(K)]
    single, batch = [], []
    want = None
    for k in range(K):
        a = _chain(counts, states[k], 1000 + k, 0xB47C4000 + k)
        a.force_stats_spec(spec)
        want = a.stats_spec()
        a.gibbs_update(n_iter)
        first = _snapshot(a)
        a.gibbs_update(3)
        single.append((first, _snapshot(a)))
        a.close()
    assert spec == 0 or want == (2 if spec == 2 or G > 8 else 4)
    ctxs = [_chain(counts, states[k], 1000 + k, 0xB47C4000 + k) for k in range(K)]
    for c in ctxs:
        c.force_stats_spec(spec)
    _lib.Context.batch_gibbs_update(ctxs, n_iter)
    for k in range(K):
        _same(single[k][0], _snapshot(ctxs[k]))
    _lib.Context.batch_gibbs_update(ctxs, 3)
    for k in range(K):
        _same(single[k][1], _snapshot(ctxs[k]))
    # a context of the batch goes on alone afterwards, on its own streams
    ctxs[0].gibbs_update(2)
    assert np.isfinite(ctxs[0].get_trace()["lp"]).all()
    for c in ctxs:
        c.close()


def test_batch_argument_checks():
    counts, _, _ = synth_counts(50, 8, 3, seed=1)
    a = _chain(counts, random_state(50, 8, 3, seed=2), 1, 2)
    b = _chain(counts[:40], random_state(40, 8, 3, seed=3), 1, 2)
    with pytest.raises(_lib.DesmanHipError):
        _lib.Context.batch_gibbs_update([a, b], 2)             # shapes differ
    a2 = _chain(counts, random_state(50, 8, 3, seed=4), 1, 2)
    a2.force_stats_spec(_lib.STATS_AGG)
    with pytest.raises(_lib.DesmanHipError):
        _lib.Context.batch_gibbs_update([a, a2], 2)            # one chain asks for another mu/E specification than the other runs
    a2.close()
    with pytest.raises(_lib.DesmanHipError):
        _lib.Context.batch_gibbs_update([a, a], 2)             # the same chain twice
    with pytest.raises(_lib.DesmanHipError):
        _lib.Context.batch_gibbs_update([a] * 9, 2)            # more than 8
    _lib.Context.batch_gibbs_update([a], 2)                    # a batch of one is allowed
    a.close(); b.close()


@pytest.mark.parametrize("V,S,G,K", [(120, 16, 4, 3), (300, 64, 8, 5), (90, 96, 12, 2), (50, 7, 1, 4), (90, 96, 3, 2), (70, 130, 4, 3),
                                     (45, 300, 8, 2), (33, 500, 12, 2)])
def test_batched_nmft_factorize_equals_one_by_one(V, S, G, K):
    """dsm_batch_nmft_factorize: same factors, update counts and objective traces as dsm_nmft_factorize per chain (the
    chains stop at different updates).  Round 5: S > 128 no longer falls back (nmft_split_kernel_b); (96, 3): the vector-ALU
    contractions of up to four haplotypes at five / six tiles."""
    counts, _, _ = synth_counts(V, S, max(G, 2), seed=700 + V)
    rs = np.random.RandomState(5)
    starts = []
    for k in range(K):
        gam0 = np.ascontiguousarray(rs.dirichlet(np.full(G, 0.3), size=S).T) if G > 1 else np.ones((1, S))
        d = rs.dirichlet(np.full(4, 0.3), size=V * G).reshape(V, G, 4)
        tau0 = np.ascontiguousarray(np.transpose(d, (2, 0, 1)).reshape(4 * V, G))
        starts.append((tau0, gam0))
    max_iter = 300
    for mc in (1e-3, 1e-7):                                  # chains that stop early at different updates / run to max_iter
        singles = []
        for tau0, gam0 in starts:
            c = _lib.Context(0); c.set_counts(counts); c.nmft_set(tau0, gam0)
            n, tr = c.nmft_factorize(max_iter, mc)
            singles.append((n, tr, c.nmft_get(), c.nmft_get_tau()))
            c.close()
        ctxs = []
        for tau0, gam0 in starts:
            c = _lib.Context(0); c.set_counts(counts); c.nmft_set(tau0, gam0)
            ctxs.append(c)
        res = _lib.Context.batch_nmft_factorize(ctxs, max_iter, mc)
        for c, (n, tr), (n1, tr1, fac1, tau1) in zip(ctxs, res, singles):
            assert n == n1 and np.array_equal(tr, tr1)
            fac = c.nmft_get()
            assert np.array_equal(fac[0], fac1[0]) and np.array_equal(fac[1], fac1[1])
            assert np.array_equal(c.nmft_get_tau(), tau1)
            c.close()
    # gamma fixed (factorize_tau): the fused pass up to 128 samples, the two-half form of the split kernel above
    singles = []
    for tau0, gam0 in starts:
        c = _lib.Context(0); c.set_counts(counts); c.nmft_set(tau0, gam0)
        n, tr = c.nmft_factorize(40, 1e-7, fix_gamma=True)
        singles.append((n, tr, c.nmft_get()))
        c.close()
    ctxs = []
    for tau0, gam0 in starts:
        c = _lib.Context(0); c.set_counts(counts); c.nmft_set(tau0, gam0)
        ctxs.append(c)
    res = _lib.Context.batch_nmft_factorize(ctxs, 40, 1e-7, fix_gamma=True)
    for c, (n, tr), (n1, tr1, fac1) in zip(ctxs, res, singles):
        assert n == n1 and np.array_equal(tr, tr1)
        fac = c.nmft_get()
        assert np.array_equal(fac[0], fac1[0]) and np.array_equal(fac[1], fac1[1])
        c.close()


@pytest.mark.parametrize("V,S,G,K,n", [(400, 24, 4, 3, 14), (128, 64, 8, 6, 9)])
def test_batched_update_tau_equals_one_by_one(V, S, G, K, n):
    """dsm_batch_update_tau (the -r path): tau-only sweeps of K chains over their own (gamma, eta) traces"""
    counts, _, _ = synth_counts(V, S, G, seed=900 + V)
    rs = np.random.RandomState(3)
    stores = [(np.ascontiguousarray(rs.dirichlet(np.ones(G), size=(n, S))),
               np.ascontiguousarray(rs.dirichlet(np.ones(4), size=(n, 4)) * 0.08 + 0.92 * np.eye(4))) for _ in range(K)]
    states = [random_state(V, S, G, seed=910 + k) for k in range(K)]
    singles = []
    for k in range(K):
        c = _chain(counts, states[k], 70 + k, 0xABC0 + k)
        c.update_tau(*stores[k])
        first = _snapshot(c)
        c.update_tau(*stores[k])
        singles.append((first, _snapshot(c)))
        c.close()
    ctxs = [_chain(counts, states[k], 70 + k, 0xABC0 + k) for k in range(K)]
    for rnd in range(2):
        _lib.Context.batch_update_tau(ctxs, [g for g, _ in stores], [e for _, e in stores])
        for k in range(K):
            _same(singles[k][rnd], _snapshot(ctxs[k]))
    for c in ctxs:
        c.close()
