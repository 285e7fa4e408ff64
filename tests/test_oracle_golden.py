"""The CPU oracle (oracle/) against the golden fixtures generated from the
imported reference (tests/golden/make_golden.py).  CPU only."""
import glob
import os

import numpy as np
import pytest

from oracle import cbind, ref_numpy as rn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_mt19937_matches_numpy_legacy_stream():
    # GSL mt19937 == init_genrand seeding; uniform = u32 / 2^32 (c_sample_tau.c:174)
    for seed in (23724839, 1, 4357, 2**31 + 5):
        bg = np.random.MT19937()
        bg._legacy_seeding(seed)
        assert np.array_equal(cbind.MT19937(seed).raw(2000), bg.random_raw(2000).astype(np.uint32))
    bg = np.random.MT19937(); bg._legacy_seeding(4357)
    assert np.array_equal(cbind.MT19937(0).raw(10), bg.random_raw(10).astype(np.uint32))
    u = cbind.MT19937(23724839).uniform(3)
    assert np.allclose(u, [0.27092583, 0.94141617, 0.95813328], atol=5e-9)


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32-10
    assert [hex(x) for x in cbind.philox4x32_10([0] * 4, [0] * 2)] == \
        ['0x6627e8d5', '0xe169c58d', '0xbc57ac4c', '0x9b00dbd8']
    assert [hex(x) for x in cbind.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2)] == \
        ['0x408f276d', '0x41c83b0e', '0xa20bc7c6', '0x6d5451fd']
    assert [hex(x) for x in cbind.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344],
                                                 [0xa4093822, 0x299f31d0])] == \
        ['0xd16cfe09', '0x94fdcceb', '0x5001e420', '0x24126ea1']


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "tau_sweep_*.npz"))))
def test_tau_sweep_against_reference_python_sampler(path):
    z = np.load(path)
    tau = z["tau_in"].copy()
    n, logp = cbind.sample_tau_u(tau, z["gamma"], z["eta"], z["counts"], z["u"], want_logp=True)
    assert np.array_equal(tau, z["tau_out"])
    # one sweep visits every (variant, haplotype) once: the per-step count (c_sample_tau.c:183) is the net difference
    assert n == int((np.argmax(z["tau_in"], 2) != np.argmax(z["tau_out"], 2)).sum())
    np.testing.assert_allclose(logp, z["logp"], rtol=1e-12, atol=1e-9)
    # and the uniforms in the fixture are the GSL-flavoured MT19937 stream
    assert np.array_equal(cbind.MT19937(int(z["mt_seed"])).uniform(len(z["u"])), z["u"])


def test_tau_sweep_global_rng_equals_explicit_uniforms():
    z = np.load(os.path.join(GOLDEN, "tau_sweep_V64_S16_G5.npz"))
    t1, t2 = z["tau_in"].copy(), z["tau_in"].copy()
    cbind.initRNG(); cbind.setRNG(77)
    n1 = cbind.sample_tau(t1, z["gamma"], z["eta"], z["counts"])
    n1b = cbind.sample_tau(t1, z["gamma"], z["eta"], z["counts"])
    cbind.freeRNG()
    u = cbind.MT19937(77).uniform(2 * 64 * 5)
    n2 = cbind.sample_tau_u(t2, z["gamma"], z["eta"], z["counts"], u[:320])
    n2b = cbind.sample_tau_u(t2, z["gamma"], z["eta"], z["counts"], u[320:])
    assert (n1, n1b) == (n2, n2b) and np.array_equal(t1, t2)


def test_loglik_logpost():
    z = np.load(os.path.join(GOLDEN, "loglik.npz"))
    for i in range(int(z["n"])):
        tau, gamma, eta, counts = (z["%s_%d" % (k, i)] for k in ("tau", "gamma", "eta", "counts"))
        idx = cbind.onehot_to_idx(tau)
        ll, lp = float(z["ll_%d" % i]), float(z["lp_%d" % i])
        assert cbind.loglik(idx, gamma, eta, counts) == pytest.approx(ll, rel=1e-13)
        assert cbind.logpost(idx, gamma, eta, counts) == pytest.approx(lp, rel=1e-13)
        assert cbind.loglik_const(counts) + (counts * np.log(np.einsum(
            'ijk,lj,km->ilm', tau, gamma, eta))).sum() == pytest.approx(ll, rel=1e-13)
        assert rn.log_likelihood(tau, gamma, eta, counts) == pytest.approx(ll, rel=1e-13)
        assert rn.log_posterior(tau, gamma, eta, counts) == pytest.approx(lp, rel=1e-13)


def test_remove_degenerate():
    z = np.load(os.path.join(GOLDEN, "degenerate.npz"))
    for i in range(int(z["n"])):
        t, g = rn.remove_degenerate(z["tau_in_%d" % i], z["gamma_in_%d" % i])
        assert t.shape[1] == int(z["G_out_%d" % i])
        assert np.array_equal(t, z["tau_out_%d" % i])
        assert np.array_equal(g, z["gamma_out_%d" % i])


def _rs_from(z):
    rs = np.random.RandomState(0)
    rs.set_state(("MT19937", z["rs_key"], int(z["rs_pos"]), 0, 0.0))
    return rs


def test_sample_mu_gamma_eta_replay_reference_stream():
    z = np.load(os.path.join(GOLDEN, "gibbs_pieces.npz"))
    rs = _rs_from(z)
    E, mu = rn.sample_mu(rs, z["tau0"], z["gamma0"], z["eta0"], z["counts"])
    assert np.array_equal(E, z["E1"]) and np.array_equal(mu, z["mu1"])
    gamma = rn.sample_gamma(rs, mu)
    assert np.array_equal(gamma, z["gamma1"])
    eta = rn.sample_eta(rs, E)
    assert np.array_equal(eta, z["eta1"])
    # one-stage law used by the product == two-stage law of the reference: E and mu
    # are consistent decompositions of the counts
    assert np.array_equal(E.sum(axis=3), z["counts"]) and np.array_equal(mu.sum(axis=3), z["counts"])


def test_gibbs_update_trajectory():
    z = np.load(os.path.join(GOLDEN, "gibbs_pieces.npz"))
    G, seed = int(z["G"]), int(z["seed"])
    counts = z["counts"]
    rs = np.random.RandomState(seed)
    gamma0, tau0 = rn.sampler_ctor_draws(rs, counts.shape[0], counts.shape[1], G)
    assert np.array_equal(gamma0, z["gamma0"]) and np.array_equal(tau0, z["tau0"])
    cbind.initRNG(); cbind.setRNG(seed)
    r = rn.gibbs_update(rs, tau0, gamma0, z["eta0"], counts, 4, cbind.sample_tau)
    cbind.freeRNG()
    np.testing.assert_allclose(r["trace"]["ll"], z["ll_store"], rtol=1e-13)
    assert np.array_equal(r["trace"]["gamma"], z["gamma_store"])
    assert np.array_equal(r["trace"]["eta"], z["eta_store"])
    assert np.array_equal(r["tau"], z["tau_final"])
    assert np.array_equal(r["star"]["tau"], z["tau_star"])
    assert r["star"]["lp"] == pytest.approx(float(z["lp_star"]), rel=1e-13)
    np.testing.assert_allclose(r["trace"]["tau_sum"] / 4.0, z["tau_mean"])
    assert -2.0 * r["trace"]["ll"].mean() == pytest.approx(float(z["mean_dev"]), rel=1e-13)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "nmft_*.npz"))))
def test_nmft(path):
    z = np.load(path)
    counts, G, seed = z["counts"], int(z["G"]), int(z["seed"])
    V, S, _ = counts.shape
    F = cbind.nmft_freq(counts)
    assert np.array_equal(F, z["F"]) and np.array_equal(rn.nmft_freq(counts), z["F"])
    # host-side init draw replays the reference's RandomState stream exactly
    tau, gam = rn.nmft_random_initialize(np.random.RandomState(seed), V, S, G)
    assert np.array_equal(tau, z["tau_raw"]) and np.array_equal(gam, z["gamma_raw"])
    # numpy restatement: same BLAS -> (near) bit-identical; C restatement: own loop order
    tn, gn = np.maximum(tau, rn.EPS), np.maximum(gam, rn.EPS)
    tc, gc = tau.copy(), gam.copy()
    cbind.nmft_adjust(tc, gc)
    assert rn.nmft_objective(F, tn, gn) == pytest.approx(float(z["div0"]), rel=1e-13)
    assert cbind.nmft_objective(F, tc, gc) == pytest.approx(float(z["div0"]), rel=1e-12)
    for it in range(1, 101):
        tn, gn = rn.nmft_update(F, tn, gn)
        tn, gn = np.maximum(tn, rn.EPS), np.maximum(gn, rn.EPS)
        cbind.nmft_update(F, tc, gc)
        cbind.nmft_adjust(tc, gc)
        if it in (1, 10, 100):
            np.testing.assert_allclose(tn, z["tau_%d" % it], rtol=1e-11, atol=1e-300)
            np.testing.assert_allclose(gn, z["gamma_%d" % it], rtol=1e-11, atol=1e-300)
            # C loop order differs from BLAS: rounding-level drift only
            np.testing.assert_allclose(tc, z["tau_%d" % it], rtol=1e-7, atol=1e-13)
            np.testing.assert_allclose(gc, z["gamma_%d" % it], rtol=1e-7, atol=1e-13)
            assert cbind.nmft_objective(F, tc, gc) == pytest.approx(float(z["div_%d" % it]), rel=1e-9)
    assert np.array_equal(rn.nmft_get_tau(tn, G), z["get_tau_100"])
    assert np.array_equal(cbind.idx_to_onehot(cbind.nmft_get_tau(tc, G)), z["get_tau_100"])
    # whole factorize() loop (max_iter = 300)
    tf, gf = tau.copy(), gam.copy()
    it, tr = cbind.nmft_factorize(F, tf, gf, max_iter=300)
    np.testing.assert_allclose(tf, z["fact_tau"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(gf, z["fact_gamma"], rtol=1e-6, atol=1e-12)
    assert tr[-1] == pytest.approx(float(z["fact_div"]), rel=1e-9)
    assert np.array_equal(cbind.idx_to_onehot(cbind.nmft_get_tau(tf, G)), z["fact_get_tau"])
    # factorize_tau with gamma fixed
    t3 = rn.nmft_random_initialize_tau(np.random.RandomState(seed + 1000), V, G)
    assert np.array_equal(t3, z["ft_tau_raw"])
    cbind.nmft_factorize_tau(F, t3, np.ascontiguousarray(z["fact_gamma"]), max_iter=50)
    np.testing.assert_allclose(t3, z["ft_tau"], rtol=1e-7, atol=1e-13)
    assert np.array_equal(cbind.idx_to_onehot(cbind.nmft_get_tau(t3, G)), z["ft_get_tau"])


def test_counter_sampler_spec_matches_expectation():
    # the product's one-stage mu/E law: sample mean over iterations ~ exact mean
    from desman_amd.synth import synth_counts, random_state
    counts, _, _ = synth_counts(40, 6, 4, seed=5)
    tau, gamma, eta = random_state(40, 6, 4, seed=6)
    idx = cbind.onehot_to_idx(tau)
    e_mu, v_mu, e_E = cbind.stats_expect(idx, gamma, eta, counts)
    n = 200
    acc_mu = np.zeros_like(e_mu); acc_E = np.zeros_like(e_E)
    for it in range(n):
        mu, E = cbind.stats_counter(idx, gamma, eta, counts, seed=99, it=it)
        assert mu.sum() == counts.sum() and E.sum() == counts.sum()
        assert np.array_equal(E.sum(axis=1), counts.sum(axis=(0, 1)).astype(np.uint64))
        acc_mu += mu; acc_E += E
    z = (acc_mu / n - e_mu) / np.sqrt(v_mu / n + 1e-12)
    assert np.abs(z).max() < 4.5
    np.testing.assert_allclose(acc_E / n, e_E, rtol=0.02, atol=3.0)


def test_dirichlet_spec_has_the_law_of_the_reference_draws():
    """orc_dirichlet_counter (the restated spec of dirichlet_kernel) against the reference-pinned
    rn.sample_gamma / rn.sample_eta fed the reference's own mu / E (gibbs_pieces.npz: E is [obs,true] and
    asymmetric): same posterior means -- in particular eta row a uses the column a of the E sums."""
    z = np.load(os.path.join(GOLDEN, "gibbs_pieces.npz"))
    mu, E = z["mu1"], z["E1"]
    sum_mu = mu.sum(axis=(0, 2)).astype(np.uint64)                # [S,G]
    esum = E.sum(axis=(0, 1)).astype(np.uint64)                   # [obs,true]
    esum[0, 1] += 40; esum[2, 3] += 25                            # make the asymmetry unmistakable
    assert not np.array_equal(esum, esum.T)
    n = 3000
    g_acc = np.zeros(sum_mu.shape); e_acc = np.zeros((4, 4)); e2 = np.zeros((4, 4))
    for it in range(n):
        g, e, rp = cbind.dirichlet_counter(sum_mu, esum, seed=77, it=it)
        g_acc += g; e_acc += e; e2 += e * e
    rs = np.random.RandomState(1)
    m = 3000
    gr = np.zeros(sum_mu.shape); er = np.zeros((4, 4))
    Efake = np.zeros((1, 1, 4, 4), dtype=np.int64); Efake[0, 0] = esum.astype(np.int64)
    mufake = np.zeros((1,) + sum_mu.shape[:1] + (1,) + sum_mu.shape[1:], dtype=np.int64)
    mufake[0, :, 0, :] = sum_mu.astype(np.int64)
    for _ in range(m):
        gr += rn.sample_gamma(rs, mufake); er += rn.sample_eta(rs, Efake)
    d = 0.1 + esum.T.astype(np.float64)
    mean = d / d.sum(axis=1, keepdims=True)
    var = mean * (1 - mean) / (d.sum(axis=1, keepdims=True) + 1)
    assert np.abs((e_acc / n - mean) / np.sqrt(var / n)).max() < 4.5
    assert np.abs((e_acc / n - er / m) / np.sqrt(var / n + var / m)).max() < 4.5
    np.testing.assert_allclose(e2 / n - (e_acc / n) ** 2, var, rtol=0.2)
    a = 0.1 + sum_mu.astype(np.float64)
    gm = a / a.sum(axis=1, keepdims=True)
    gv = gm * (1 - gm) / (a.sum(axis=1, keepdims=True) + 1)
    assert np.abs((g_acc / n - gr / m) / np.sqrt(gv / n + gv / m + 1e-14)).max() < 4.5
    # rowprior = Dirichlet log-prior of each row (Desman_Utils.py:35-44)
    g, e, rp = cbind.dirichlet_counter(sum_mu, esum, seed=77, it=0)
    S, G = g.shape
    assert rp.sum() + 0.0 == pytest.approx(cbind.logprior(g, e, 0), rel=1e-12)


# ---------------------------------------------------------------- f4: accessory genes
@pytest.mark.parametrize("name", ["gene_assign", "gene_assign_lowcov"])
def test_gene_oracle_reproduces_reference_run(name):
    """oracle/ref_genes.py driven in GeneAssign.main's order reproduces the imported reference classes:
    KL start, per-gene NMFT + first sweep, two update() rounds (copy numbers, tau, per-gene ll), calcTauStar."""
    from oracle import ref_genes as rg
    from _genes_util import load_case, split
    k = load_case(name)
    z, C, G = k['z'], k['C'], k['G']
    rs = np.random.RandomState(k['seed'])
    cbind.initRNG(); cbind.setRNG(k['seed'])
    eta0 = rs.uniform(0, 1.0, (C, G))                                  # KLAssign.random_initialize
    kl_eta, n_kl, kl_div = rg.kl_assign(k['cov'], k['delta'], eta0)
    np.testing.assert_allclose(kl_eta, z['kl_eta'], rtol=1e-9, atol=1e-12)
    assert abs(kl_div - float(z['kl_div'])) < 1e-8
    eta = np.rint(kl_eta)
    eta[eta > 1.0] = 1.0                                              # Eta_Sampler.__init__:121-122 (max_eta = 2)
    np.testing.assert_array_equal(eta, z['eta_init'])
    eta = eta.astype(np.int64)
    prior = rg.eta_log_prior(2, 0.01)
    np.testing.assert_allclose(prior, z['eta_log_prior'], rtol=0, atol=1e-15)
    taus = []
    for c in range(C):                                               # __init__:126-135
        x = k['variants'][c]
        if x.shape[0] == 0:
            taus.append(np.zeros((0, G, 4), dtype=np.int64)); continue
        gr = rg.mask_gamma(k['gamma'], eta[c])
        t, _ = rg.gene_nmft_tau(rs, x, gr, G)
        cbind.sample_tau(t, np.ascontiguousarray(gr), k['eps'], x)
        taus.append(t)
    np.testing.assert_array_equal(np.concatenate(taus), z['tau_init'])
    eta_star = np.zeros_like(eta); llstar = np.zeros(C)
    for rnd in (1, 2):
        store, trace, gene_ll = rg.eta_update_reference_order(rs, eta, taus, k['variants'], k['cov'], k['gamma'], k['eps'],
                                                              k['delta_gs'], prior, k['iters'], eta_star, llstar)
        np.testing.assert_array_equal(store, z['eta_store_%d' % rnd])
        np.testing.assert_array_equal(eta_star, z['eta_star_%d' % rnd])
        np.testing.assert_array_equal(np.concatenate(taus), z['tau_%d' % rnd])
        np.testing.assert_allclose(gene_ll, z['gene_ll_%d' % rnd], rtol=1e-11)
        np.testing.assert_allclose(llstar, z['gene_llstar_%d' % rnd], rtol=1e-11)
        np.testing.assert_allclose(trace, z['ll_log_%d' % rnd], atol=2e-6)      # the log line prints 6 decimals
    stars, lls, stores = rg.calc_tau_star(rs, eta, eta_star, k['variants'], k['gamma'], k['eps'], k['tau_iter'], G)
    np.testing.assert_array_equal(np.concatenate(stars), z['tau_star'])
    np.testing.assert_array_equal(np.concatenate(stores, axis=1), z['tau_store'])
    ref_ll = z['tau_star_ll']
    mine = np.concatenate(lls)
    live = ref_ll > -1e300
    np.testing.assert_array_equal(live, mine > -1e300)
    np.testing.assert_allclose(mine[live], ref_ll[live], rtol=1e-11)
    # the same call with a substitute gamma / epsilon (Eta_Sampler.py:397-403), continuing both streams
    stars, lls, stores = rg.calc_tau_star(rs, eta, eta_star, k['variants'], k['gamma'], k['eps'], k['tau_iter'], G,
                                          gamma_sub=z['sub_gamma'], eps_sub=z['sub_epsilon'])
    np.testing.assert_array_equal(np.concatenate(stars), z['sub_tau_star'])
    np.testing.assert_array_equal(np.concatenate(stores, axis=1), z['sub_tau_store'])
    ref_ll, mine = z['sub_tau_star_ll'], np.concatenate(lls)
    live = ref_ll > -1e300
    np.testing.assert_allclose(mine[live], ref_ll[live], rtol=1e-11)


def test_masked_row_sum_order_is_numpys():
    """the device and the oracle re-normalise the masked gamma with numpy's add.reduce order (sequential below
    8 haplotypes, 8 running sums above): restated here and compared with ndarray.sum bit for bit."""
    def row_sum(a):
        n = len(a)
        if n < 8:
            r = 0.0
            for x in a:
                r += x
            return r
        r = [a[j] for j in range(8)]
        i = 8
        while i < n - (n % 8):
            for j in range(8):
                r[j] += a[i + j]
            i += 8
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        while i < n:
            res += a[i]
            i += 1
        return res
    rng = np.random.default_rng(5)
    for G in range(1, 33):
        A = rng.random((9, G)) * rng.random((9, 1))
        A[:, rng.random(G) < 0.3] = 0.0
        np.testing.assert_array_equal(A.sum(axis=1), np.array([row_sum(list(r)) for r in A]))
