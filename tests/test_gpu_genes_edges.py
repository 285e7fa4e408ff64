"""Edge shapes of the accessory-gene path against the CPU oracle (which tests/test_oracle_golden.py pins to the
reference): more than 8 haplotypes (numpy's blocked row sums in maskGamma), more than 64 samples (two samples
per lane), genes larger than one workgroup pass, three copy-number states, no variants at all, variant
subsampling + restoreFullVariants."""
import numpy as np
import pandas as pd
import pytest
from scipy.special import gammaln

from desman_amd.synth import synth_genes

pytestmark = pytest.mark.gpu


def _case(C, S, G, vmax, seed, **kw):
    d = synth_genes(C, S, G, seed=seed, vmax=vmax, **kw)
    gamma = np.ascontiguousarray(d['gamma'])
    delta = np.ascontiguousarray(gamma * d['total_mean'][:, None])
    off = np.concatenate([[0], np.cumsum(np.bincount(d['gene_of'], minlength=C))]).astype(np.int32)
    variants = [np.ascontiguousarray(d['counts'][off[c]:off[c + 1]]) for c in range(C)]
    return dict(d=d, C=C, S=S, G=G, gamma=gamma, delta=delta, delta_gs=np.ascontiguousarray(delta.T), gene_off=off,
                variants=variants, eps=np.ascontiguousarray(d['epsilon']), cov=np.ascontiguousarray(d['cov']))


def _device(k, eta, tau, max_eta):
    from desman_amd import _lib
    from oracle import ref_genes as rg
    dev = _lib.Genes(0)
    x = k['d']['counts']
    dev.set_data(x, k['gene_off'], k['cov'])
    per_v = (gammaln(x.sum(axis=2) + 1.0) - gammaln(x + 1.0).sum(axis=2)).sum(axis=1) if len(x) else np.zeros(0)
    mult = np.array([per_v[k['gene_off'][c]:k['gene_off'][c + 1]].sum() for c in range(k['C'])])
    prior = rg.eta_log_prior(max_eta, 0.01)
    dev.set_model(k['gamma'], k['eps'], k['delta_gs'], max_eta, prior, -gammaln(k['cov'] + 1.0).sum(axis=1), mult)
    dev.set_state(eta.astype(np.int32), tau)
    dev.seed(3)
    return dev, prior


@pytest.mark.parametrize("C,S,G,vmax,max_eta,kw", [
    (10, 70, 9, 40, 3, dict(mean_lo=0.5, mean_hi=3.0)),      # blocked row sums, 2 samples / lane, multi-workgroup genes
    (6, 5, 2, 3, 2, dict(mean_lo=0.3, mean_hi=1.5)),         # 16-lane groups
    (8, 24, 5, 90, 2, dict(mean_lo=1.0, mean_hi=4.0)),       # 32-lane groups, a gene of > 64 variants
    (4, 130, 12, 6, 2, dict(mean_lo=0.5, mean_hi=2.0)),      # 3 samples / lane
])
def test_batched_update_edge_shapes(C, S, G, vmax, max_eta, kw):
    from oracle import ref_genes as rg
    from oracle import cbind
    k = _case(C, S, G, vmax, seed=C * 100 + G, **kw)
    rng = np.random.default_rng(S)
    Vtot = int(k['gene_off'][-1])
    eta0 = rng.integers(0, max_eta, size=(C, G))
    eta0[0, :] = 0                                           # a gene currently in no haplotype
    eta0[1, :] = 0; eta0[1, 0] = 1                           # a gene in exactly one
    tau0 = cbind.idx_to_onehot(rng.integers(0, 4, size=(Vtot, G)))
    n_iter = 3
    u_tau = rng.integers(0, 2 ** 32, size=(n_iter, G, 2, Vtot * G), dtype=np.uint32)
    u_eta = rng.random((n_iter, C, G))
    dev, prior = _device(k, eta0, tau0, max_eta)
    store, trace = dev.update(n_iter, reset_star=True, u_tau_ext=u_tau, u_eta_ext=u_eta)
    eta_dev, tau_dev = dev.get_state()
    eta = eta0.copy()
    taus = [np.ascontiguousarray(tau0[k['gene_off'][c]:k['gene_off'][c + 1]]) for c in range(C)]
    eta_star = np.zeros_like(eta); llstar = np.zeros(C)
    ref_store, ref_trace = rg.eta_update_batched(eta, taus, k['variants'], k['gene_off'], k['cov'], k['gamma'], k['eps'],
                                                 k['delta_gs'], prior, n_iter, u_tau, u_eta, eta_star, llstar)
    np.testing.assert_array_equal(store, ref_store)
    np.testing.assert_array_equal(tau_dev, np.concatenate(taus))
    np.testing.assert_allclose(trace, ref_trace, rtol=1e-10)
    star_dev, llstar_dev = dev.get_star()
    np.testing.assert_array_equal(star_dev, eta_star)


def _frames(k):
    S = k['S']
    samples = ["s%02d" % s for s in range(S)]
    cov = pd.DataFrame(k['cov'], index=k['d']['genes'], columns=samples)
    cols = [s + "-" + b for s in samples for b in "ACGT"]
    x = k['d']['counts']
    var = pd.DataFrame(x.reshape(x.shape[0], S * 4), index=[k['d']['genes'][c] for c in k['d']['gene_of']], columns=cols)
    return cov, var


def test_subsample_then_restore_full_variants():
    """max_var < V_c: the constructor keeps a sorted random subset (RandomState.choice per gene, in gene order),
    update() runs on it, restoreFullVariants + calcTauStar use every row again -- all against the oracle."""
    from desman_amd import sampletau
    from desman_amd.Eta_Sampler import Eta_Sampler
    from oracle import ref_genes as rg
    from oracle import cbind
    C, S, G, max_var, seed, iters, tau_iter = 6, 9, 3, 4, 31, 3, 2
    k = _case(C, S, G, 11, seed=8, mean_lo=0.5, mean_hi=3.0)
    cov, var = _frames(k)
    init = (np.random.default_rng(1).random((C, G)) < 0.6).astype(float)
    init[init.sum(axis=1) == 0, 0] = 1.0
    sampletau.initRNG(); sampletau.setRNG(seed)
    smp = Eta_Sampler(np.random.RandomState(seed), var, cov, k['gamma'], k['delta'], np.ones(S), k['eps'], init,
                      max_iter=iters, tau_iter=tau_iter, max_var=max_var)
    # ---- oracle, same order of draws
    rs = np.random.RandomState(seed)
    cbind.initRNG(); cbind.setRNG(seed)
    full = k['variants']
    sub = []
    for c in range(C):
        nv = full[c].shape[0]
        sub.append(full[c][np.sort(rs.choice(nv, max_var, replace=False))] if nv > max_var else full[c])
    assert any(s.shape[0] < f.shape[0] for s, f in zip(sub, full))
    eta = init.astype(np.int64)
    taus = []
    for c in range(C):
        if sub[c].shape[0] == 0:
            taus.append(np.zeros((0, G, 4), dtype=np.int64)); continue
        gr = rg.mask_gamma(k['gamma'], eta[c])
        t, _ = rg.gene_nmft_tau(rs, np.ascontiguousarray(sub[c]), gr, G)
        cbind.sample_tau(t, np.ascontiguousarray(gr), k['eps'], np.ascontiguousarray(sub[c]))
        taus.append(t)
    np.testing.assert_array_equal(smp._tau, np.concatenate(taus))
    prior = rg.eta_log_prior(2, 0.01)
    eta_star = np.zeros_like(eta); llstar = np.zeros(C)
    sub = [np.ascontiguousarray(s) for s in sub]
    store, trace, gene_ll = rg.eta_update_reference_order(rs, eta, taus, sub, k['cov'], k['gamma'], k['eps'], k['delta_gs'],
                                                          prior, iters, eta_star, llstar)
    smp.update()
    np.testing.assert_array_equal(smp.eta_store, store)
    np.testing.assert_array_equal(smp._tau, np.concatenate(taus))
    np.testing.assert_allclose(smp.gene_ll, gene_ll, rtol=1e-10)
    smp.restoreFullVariants()
    assert smp._Vtot == sum(f.shape[0] for f in full)
    smp.calcTauStar(smp.eta_star)
    stars, lls, stores = rg.calc_tau_star(rs, eta, eta_star, full, k['gamma'], k['eps'], tau_iter, G)
    np.testing.assert_array_equal(smp._tau_star_cat, np.concatenate(stars))
    np.testing.assert_array_equal(smp._tau_store_cat, np.concatenate(stores, axis=1))


def test_no_variant_table_at_all():
    """GeneAssign without -v: every gene is decided by its coverage alone; both samplers run, the exact one
    equals the oracle."""
    from desman_amd import sampletau
    from desman_amd.Eta_Sampler import Eta_Sampler
    from oracle import ref_genes as rg
    C, S, G, seed, iters = 9, 7, 4, 77, 4
    k = _case(C, S, G, 5, seed=3, mean_lo=0.3, mean_hi=2.0)
    cov, _ = _frames(k)
    init = np.ones((C, G))
    sampletau.initRNG(); sampletau.setRNG(seed)
    smp = Eta_Sampler(np.random.RandomState(seed), None, cov, k['gamma'], k['delta'], np.ones(S), k['eps'], init, max_iter=iters)
    smp.update()
    rs = np.random.RandomState(seed)
    eta = init.astype(np.int64)
    none = [np.zeros((0, S, 4), dtype=np.int64)] * C
    taus = [np.zeros((0, G, 4), dtype=np.int64) for _ in range(C)]
    eta_star = np.zeros_like(eta); llstar = np.zeros(C)
    store, trace, gene_ll = rg.eta_update_reference_order(rs, eta, taus, none, k['cov'], k['gamma'], k['eps'], k['delta_gs'],
                                                          rg.eta_log_prior(2, 0.01), iters, eta_star, llstar)
    np.testing.assert_array_equal(smp.eta_store, store)
    np.testing.assert_allclose(smp.gene_ll, gene_ll, rtol=1e-11)
    np.testing.assert_array_equal(smp.eta_star, eta_star)
    fast = Eta_Sampler(np.random.RandomState(seed), None, cov, k['gamma'], k['delta'], np.ones(S), k['eps'], init,
                       max_iter=50, rng="philox")
    fast.update()
    assert fast.eta_store.shape == (50, C, G) and np.isfinite(fast.gene_ll).all()
    fast.calcTauStar(fast.eta_star)
    assert fast.getTauStar(None)[0].shape == (0, G, 4)


def test_gene_api_rejects_bad_input():
    from desman_amd import _lib
    dev = _lib.Genes(0)
    with pytest.raises(_lib.DesmanHipError):
        dev.loglik()                                                  # no data yet
    cov = np.ones((2, 3))
    with pytest.raises(_lib.DesmanHipError):
        dev.set_data(np.zeros((4, 3, 4), dtype=np.int64), np.array([0, 5, 4], dtype=np.int32), cov)   # offsets not monotone
    dev.set_data(np.zeros((4, 3, 4), dtype=np.int64), np.array([0, 1, 4], dtype=np.int32), cov)
    with pytest.raises(_lib.DesmanHipError):
        dev.set_state(np.zeros((2, 2), dtype=np.int32), None)         # no model yet
    g = np.full((3, 2), 0.5)
    dev.set_model(g, np.eye(4) * 0.96 + 0.01, np.ones((2, 3)), 2, np.array([-0.01, -4.6]), np.zeros(2), np.zeros(2))
    with pytest.raises(_lib.DesmanHipError):
        dev.set_state(np.full((2, 2), 5, dtype=np.int32), None)       # copy number outside 0..max_eta-1
    with pytest.raises(_lib.DesmanHipError):
        dev.set_data(np.full((4, 3, 4), -1, dtype=np.int64), np.array([0, 1, 4], dtype=np.int32), cov)  # negative count


def test_philox_sampler_is_invariant_to_gene_sharding():
    """rng='philox': every draw is keyed by (global gene, row within the gene, haplotype, iteration), so two
    samplers over the two halves of the genes (gene_base = first gene of the shard) reproduce the single sampler
    over all genes: NMFT start, first sweep, update() trajectory, MAP record, calcTauStar -- with subsampling."""
    from desman_amd import sampletau
    from desman_amd.Eta_Sampler import Eta_Sampler
    from desman_amd.gene_shards import partition_genes
    C, S, G, seed, iters = 14, 10, 4, 5, 6
    k = _case(C, S, G, 9, seed=21, mean_lo=0.4, mean_hi=2.5)
    cov, var = _frames(k)
    init = (np.random.default_rng(2).random((C, G)) < 0.6).astype(float)
    init[init.sum(axis=1) == 0, 1] = 1.0
    sampletau.initRNG(); sampletau.setRNG(seed)

    def run(lo, hi):
        names = k['d']['genes'][lo:hi]
        sub = var[var.index.isin(set(names))]
        s = Eta_Sampler(np.random.RandomState(seed), sub, cov.iloc[lo:hi], k['gamma'], k['delta'], np.ones(S), k['eps'],
                        init[lo:hi], max_iter=iters, tau_iter=2, max_var=5, rng="philox", gene_base=lo)
        tau0 = s._tau.copy()
        s.update()
        s.restoreFullVariants()
        s.calcTauStar(s.eta_star)
        return tau0, s.eta_store.copy(), s.eta_star.copy(), s.gene_llstar.copy(), s._tau_star_cat.copy(), s._tau_store_cat.copy()

    whole = run(0, C)
    b = partition_genes(np.bincount(k['d']['gene_of'], minlength=C), 2)
    parts = [run(int(b[r]), int(b[r + 1])) for r in range(2)]
    assert 0 < b[1] < C
    np.testing.assert_array_equal(whole[0], np.concatenate([p[0] for p in parts]))
    np.testing.assert_array_equal(whole[1], np.concatenate([p[1] for p in parts], axis=1))
    np.testing.assert_array_equal(whole[2], np.concatenate([p[2] for p in parts]))
    np.testing.assert_allclose(whole[3], np.concatenate([p[3] for p in parts]), rtol=1e-13)
    np.testing.assert_array_equal(whole[4], np.concatenate([p[4] for p in parts]))
    np.testing.assert_array_equal(whole[5], np.concatenate([p[5] for p in parts], axis=1))
    assert whole[1].std() > 0                                            # the chain moved
