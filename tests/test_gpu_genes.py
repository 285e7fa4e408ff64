"""GPU parity of the accessory-gene path: desman_amd.Eta_Sampler / GeneAssign and the
dsm_genes_* / dsm_kl_assign entry points against (a) the states the reference classes produced
(tests/golden/gene_assign*.npz) and (b) the CPU oracle (oracle/ref_genes.py) on the same inputs."""
import os

import numpy as np
import pandas as pd
import pytest

from _genes_util import load_case, split

pytestmark = pytest.mark.gpu

CASES = ["gene_assign", "gene_assign_lowcov"]


def _frames(k, tmp_path):
    from desman_amd.synth import write_gene_inputs
    os.makedirs(str(tmp_path), exist_ok=True)
    paths = write_gene_inputs(k['d'], str(tmp_path))
    read = lambda p: pd.read_csv(p, header=0, index_col=0)
    scg, gam, cov, var = read(paths[0]), read(paths[1]), read(paths[2]), read(paths[4])
    names = sorted(set(gam.index.values) & set(scg.index.values) & set(cov.columns.values))
    return paths, scg.reindex(names), cov[names], var, names


def _sampler(k, tmp_path, rng="mt19937", **kw):
    from desman_amd import sampletau
    from desman_amd.Eta_Sampler import Eta_Sampler
    from desman_amd.GeneAssign import KLAssign, expand_sample_names
    paths, scg, cov, var, names = _frames(k, tmp_path)
    prng = np.random.RandomState(k['seed'])
    sampletau.initRNG(); sampletau.setRNG(k['seed'])
    kl = KLAssign(prng, cov.to_numpy(), k['delta'])
    kl.factorize()
    smp = Eta_Sampler(prng, var[expand_sample_names(names)], cov, k['gamma'], k['delta'], scg['sd'].to_numpy(), k['eps'],
                      np.rint(kl.eta), max_iter=k['iters'], max_eta=2, max_var=int(1e10), tau_iter=k['tau_iter'], rng=rng, **kw)
    return kl, smp, var


@pytest.mark.parametrize("name", CASES)
def test_eta_sampler_reproduces_reference_run(name, tmp_path):
    """rng='mt19937': KL start, NMFT + first sweep, two update() rounds and calcTauStar equal the imported
    reference classes' states: copy numbers, tau and MAP record exactly, log-likelihoods to 1e-10."""
    k = load_case(name)
    z = k['z']
    kl, smp, var = _sampler(k, tmp_path)
    np.testing.assert_allclose(kl.eta, z['kl_eta'], rtol=1e-8, atol=1e-11)
    np.testing.assert_array_equal(smp.eta, z['eta_init'])
    np.testing.assert_array_equal(smp._tau, z['tau_init'])
    np.testing.assert_allclose(smp.eta_log_prior, z['eta_log_prior'], rtol=0, atol=1e-15)
    for rnd in (1, 2):
        smp.update()
        np.testing.assert_array_equal(smp.eta_store, z['eta_store_%d' % rnd])
        np.testing.assert_array_equal(smp.eta_star, z['eta_star_%d' % rnd])
        np.testing.assert_array_equal(smp._tau, z['tau_%d' % rnd])
        np.testing.assert_allclose(smp.gene_ll, z['gene_ll_%d' % rnd], rtol=1e-10)
        np.testing.assert_allclose(smp.gene_llstar, z['gene_llstar_%d' % rnd], rtol=1e-10)
        assert abs(smp.ll - float(z['ll_%d' % rnd])) < 1e-8 * abs(float(z['ll_%d' % rnd]))
    smp.restoreFullVariants()
    smp.calcTauStar(smp.eta_star)
    tau_star, tau_mean, pos, owner = smp.getTauStar(var)
    np.testing.assert_array_equal(tau_star, z['tau_star'])
    np.testing.assert_array_equal(tau_mean, z['tau_mean_trunc'])
    np.testing.assert_array_equal(smp._tau_store_cat, z['tau_store'])
    np.testing.assert_array_equal(pos, z['pos'])
    assert list(owner) == list(z['contig_index'])
    ref = z['tau_star_ll']
    mine = np.concatenate([smp.gene_ll_tau_star[g] for g in smp.genes])
    live = ref > -1e300
    np.testing.assert_array_equal(live, mine > -1e300)
    np.testing.assert_allclose(mine[live], ref[live], rtol=1e-10)
    # calcTauStar with a substitute gamma / epsilon (Eta_Sampler.py:397-403; used by the sweeps and their likelihoods only):
    # continues both streams from the first call, as the run that wrote the fixture did
    smp.calcTauStar(smp.eta_star, gamma=z['sub_gamma'], epsilon=z['sub_epsilon'])
    tau_star2, _, _, _ = smp.getTauStar(var)
    np.testing.assert_array_equal(tau_star2, z['sub_tau_star'])
    np.testing.assert_array_equal(smp._tau_store_cat, z['sub_tau_store'])
    ref, mine = z['sub_tau_star_ll'], np.concatenate([smp.gene_ll_tau_star[g] for g in smp.genes])
    live = ref > -1e300
    np.testing.assert_array_equal(live, mine > -1e300)
    np.testing.assert_allclose(mine[live], ref[live], rtol=1e-10)
    assert (tau_star2 != tau_star).any()                      # the substitute model does change the outcome
    # ... and leaves the sampler's own model in place
    np.testing.assert_array_equal(smp.gamma, k['gamma'])
    with pytest.raises(ValueError):
        smp.calcTauStar(smp.eta_star, gamma=z['sub_gamma'][:, :-1])


@pytest.mark.parametrize("name", CASES)
def test_geneassign_cli_writes_reference_files(name, tmp_path):
    """python -m desman_amd.GeneAssign with the reference's arguments: the six output files against the ones
    the reference's main() wrote (integer-valued tables byte for byte, the KL table numerically)."""
    import io
    from desman_amd import GeneAssign
    k = load_case(name)
    z = k['z']
    paths, *_ = _frames(k, tmp_path)
    stub = os.path.join(str(tmp_path), "ga")
    GeneAssign.main([paths[0], paths[1], paths[2], paths[3], "-s", str(k['seed']), "-i", str(k['iters']), "-o", stub,
                     "-v", paths[4], "--assign_tau"])
    for suffix in ("etaD_df.csv", "etaS_df.csv", "etaM_df.csv", "_tau_star.csv", "_tau_mean.csv"):
        with open(stub + suffix) as fh:
            assert fh.read() == str(z['file_' + suffix.replace('.csv', '').strip('_')]), suffix
    mine = pd.read_csv(stub + "eta_df.csv", index_col=0)
    ref = pd.read_csv(io.StringIO(str(z['file_eta_df'])), index_col=0)
    assert list(mine.index) == list(ref.index)
    np.testing.assert_allclose(mine.to_numpy(), ref.to_numpy(), rtol=1e-8, atol=1e-11)


def _genes_from_golden(k, eta, tau):
    from scipy.special import gammaln
    from desman_amd import _lib
    from oracle import ref_genes as rg
    dev = _lib.Genes(0)
    counts = k['d']['counts']
    dev.set_data(counts, k['gene_off'], k['cov'])
    per_v = (gammaln(counts.sum(axis=2) + 1.0) - gammaln(counts + 1.0).sum(axis=2)).sum(axis=1)
    mult = np.array([per_v[k['gene_off'][c]:k['gene_off'][c + 1]].sum() for c in range(k['C'])])
    prior = rg.eta_log_prior(2, 0.01)
    dev.set_model(k['gamma'], k['eps'], k['delta_gs'], 2, prior, -gammaln(k['cov'] + 1.0).sum(axis=1), mult)
    dev.set_state(eta.astype(np.int32), tau.astype(np.int64))
    dev.seed(1)
    return dev, prior


@pytest.mark.parametrize("name", CASES)
def test_batched_update_equals_spec_with_explicit_uniforms(name):
    """dsm_genes_update fed explicit uniforms == oracle.ref_genes.eta_update_batched on the same uniforms:
    copy-number trajectory, final tau and MAP record exactly, per-gene log-likelihood trace to 1e-10."""
    from oracle import ref_genes as rg
    k = load_case(name)
    z, C, G = k['z'], k['C'], k['G']
    eta0, tau0 = z['eta_init'].astype(np.int64), z['tau_init'].astype(np.int64)
    Vtot, n_iter = tau0.shape[0], 5
    rng = np.random.default_rng(99)
    u_tau = rng.integers(0, 2 ** 32, size=(n_iter, G, 2, Vtot * G), dtype=np.uint32)
    u_eta = rng.random((n_iter, C, G))
    dev, prior = _genes_from_golden(k, eta0, tau0)
    store, trace = dev.update(n_iter, reset_star=True, u_tau_ext=u_tau, u_eta_ext=u_eta)
    eta_dev, tau_dev = dev.get_state()
    star_dev, llstar_dev = dev.get_star()
    eta = eta0.copy()
    taus = split(tau0, k['gene_off'])
    eta_star = np.zeros_like(eta); llstar = np.zeros(C)
    ref_store, ref_trace = rg.eta_update_batched(eta, taus, k['variants'], k['gene_off'], k['cov'], k['gamma'], k['eps'],
                                                 k['delta_gs'], prior, n_iter, u_tau, u_eta, eta_star, llstar)
    np.testing.assert_array_equal(store, ref_store)
    np.testing.assert_array_equal(eta_dev, eta)
    np.testing.assert_array_equal(tau_dev, np.concatenate(taus))
    np.testing.assert_allclose(trace, ref_trace, rtol=1e-10)
    np.testing.assert_array_equal(star_dev, eta_star)
    np.testing.assert_allclose(llstar_dev, llstar, rtol=1e-10)


def test_batched_sampler_has_the_exact_samplers_law(tmp_path):
    """rng='philox' against rng='mt19937' on the flat-posterior case: per (gene, haplotype) posterior mean of the
    copy number over long runs agree within Monte-Carlo error (the two samplers share every conditional)."""
    k = load_case("gene_assign_lowcov")
    k['iters'] = 3000
    _, exact, _ = _sampler(k, tmp_path / "a")
    exact.update()
    _, fast, _ = _sampler(k, tmp_path / "b", rng="philox")
    fast.update()
    burn, nb = 200, 14

    def mean_and_se(store):                       # batch means: the chains are autocorrelated
        blocks = store[burn:].reshape(nb, -1, *store.shape[1:]).mean(axis=1)
        return blocks.mean(axis=0), blocks.std(axis=0, ddof=1) / np.sqrt(nb)

    m1, s1 = mean_and_se(exact.eta_store)
    m2, s2 = mean_and_se(fast.eta_store)
    z = np.abs(m1 - m2) / np.sqrt(s1 ** 2 + s2 ** 2 + 1e-4)
    assert z.max() < 5.0, (z.max(), m1, m2)
    assert abs(m1.mean() - m2.mean()) < 0.03


@pytest.mark.parametrize("C,S,G", [(7, 8, 3), (300, 21, 9), (33, 70, 4)])
def test_kl_assign_matches_oracle(C, S, G):
    from desman_amd import _lib
    from oracle import ref_genes as rg
    rng = np.random.default_rng(C)
    delta = rng.random((S, G)) * 50.0
    truth = (rng.random((C, G)) < 0.5).astype(float)
    cov = rng.poisson(truth @ delta.T + 0.3).astype(float)
    cov[0, :] = 0.0                                                   # a gene nobody covers
    eta0 = rng.random((C, G))
    eta, n, div = _lib.kl_assign(cov, delta, eta0, max_iter=2000)
    ref_eta, ref_n, ref_div = rg.kl_assign(cov, delta, eta0, max_iter=2000)
    assert abs(n - ref_n) <= 2, (n, ref_n)
    assert abs(div - ref_div) <= 1e-6 * max(1.0, abs(ref_div))
    np.testing.assert_allclose(eta, ref_eta, rtol=1e-4, atol=1e-6)
    np.testing.assert_array_equal(np.rint(eta), np.rint(ref_eta))


@pytest.mark.parametrize("name", CASES)
def test_gene_nmft_start_matches_oracle(name):
    """per-gene factorize_tau with the masked gamma: arg-max tau of every gene equals the C oracle's, update
    counts within one (the sums over samples keep the oracle's order; only log() differs in the last bit)"""
    from oracle import ref_genes as rg
    k = load_case(name)
    z, C, G = k['z'], k['C'], k['G']
    eta0 = z['eta_init'].astype(np.int64)
    dev, _ = _genes_from_golden(k, eta0, np.zeros_like(z['tau_init'], dtype=np.int64))
    rs_dev, rs_ref = np.random.RandomState(5), np.random.RandomState(5)
    Vtot = z['tau_init'].shape[0]
    start = np.full((Vtot, 4, G), 0.25)
    for c in range(C):
        lo, hi = k['gene_off'][c], k['gene_off'][c + 1]
        if hi > lo:
            d = rs_dev.dirichlet(np.full(4, 0.01), size=(hi - lo) * G).reshape(hi - lo, G, 4)
            start[lo:hi] = np.transpose(d, (0, 2, 1))
    n_dev = dev.nmft_tau(start)
    _, tau_dev = dev.get_state()
    for c in range(C):
        lo, hi = k['gene_off'][c], k['gene_off'][c + 1]
        if hi == lo:
            assert n_dev[c] == -1
            continue
        t, n = rg.gene_nmft_tau(rs_ref, k['variants'][c], rg.mask_gamma(k['gamma'], eta0[c]), G)
        np.testing.assert_array_equal(tau_dev[lo:hi], t)
        assert abs(int(n_dev[c]) - n) <= 1, (c, n_dev[c], n)


def test_geneassign_cli_batched_sampler(tmp_path):
    """--rng philox: the same six files with the same layout (the chain itself is a different realisation: on this
    multimodal case it leaves the KL start the reference-stream chain stays in)."""
    import io
    from desman_amd import GeneAssign
    k = load_case("gene_assign")
    z = k['z']
    paths, *_ = _frames(k, tmp_path)
    stub = os.path.join(str(tmp_path), "gb")
    known = os.path.join(str(tmp_path), "genomes.csv")
    pd.DataFrame(k['d']['eta_true'], index=k['d']['genes']).to_csv(known)
    GeneAssign.main([paths[0], paths[1], paths[2], paths[3], "-s", str(k['seed']), "-i", "10", "-o", stub, "-v", paths[4],
                     "--assign_tau", "--rng", "philox", "-g", known])
    for suffix in ("etaD_df.csv", "etaS_df.csv", "etaM_df.csv", "eta_df.csv", "_tau_star.csv", "_tau_mean.csv"):
        assert os.path.exists(stub + suffix), suffix
    eta_s = pd.read_csv(stub + "etaS_df.csv", index_col=0)
    assert list(eta_s.index) == k['d']['genes'] and set(np.unique(eta_s.to_numpy())) <= {0.0, 1.0}
    ts = pd.read_csv(stub + "_tau_star.csv", index_col=0)
    ref = pd.read_csv(io.StringIO(str(z['file_tau_star'])), index_col=0)
    assert list(ts.index) == list(ref.index) and list(ts['Position']) == list(ref['Position'])
    t = ts.to_numpy()[:, 1:].reshape(len(ts), k['G'], 4)
    carried = np.repeat(eta_s.to_numpy(), np.diff(k['gene_off']), axis=0) > 0       # [row, haplotype]
    assert (t.sum(axis=2) == 1)[carried].all()                                      # one base per carried haplotype
