"""Law of the counter-based mu/E specification (oracle/stats_agg.c, orc_stats_counter), on the CPU.

The HIP kernels reproduce these functions variate by variate (tests/test_gpu_parity.py:
test_binomial_and_multinomial_samplers_match_spec, test_stats_*), so pinning the LAW of the specification once, here, pins
the law of the device draws.  What it is pinned to:
  * the exact binomial / multinomial pmf (chi-square, >= 2e5 variates) for every sampler the specification is built from --
    sequential-search inversion, Hoermann's BTRS, the read-by-read draw, mult4 -- on both sides of every switch between them;
  * the exact law of stage 2's halving tree (sums of independent binomials) from hand-built subset counts;
  * the reference's own sampleMu (/root/reference/desman/HaploSNP_Sampler.py:284-309, restated RandomState-exactly in
    oracle/ref_numpy.py and pinned by golden fixtures): two-sample chi-square on every marginal of sum_mu and Esum at
    G = 10 / 12, x 15 depth and a converged state with eta ~ 0.97 I.
"""
import numpy as np
import pytest
from scipy import stats as st

from oracle import cbind

from _law import (LAW_CASES, assert_same_law, chi2_vs_binom, chi2_vs_pmf, law_case, reference_draws, sum_of_binomials_pmf)

NV = 200000


# (kind, n, p): kind 0 = stage-1 binomial (inversion while n q <= 128, BTRS above), kind 1 = stage-2 binomial (switch at 16).
# Both tails (p > 1/2 takes the flipped branch), means 0.3 ... 1e6, either side of each switch, counts up to 2^32 - 1.
BINOM_CASES = [
    (0, 1000, 0.0003), (0, 1000, 0.005), (0, 40, 0.4), (0, 1000, 0.016), (0, 1000, 0.064), (0, 300, 0.425),
    (0, 1000, 0.1279), (0, 1000, 0.1281), (0, 260, 0.4923), (0, 260, 0.4925),          # n q = 127.9 / 128.1 / 127.998 / 128.05
    (0, 1000, 0.9997), (0, 1000, 0.984), (0, 1000, 0.8721), (0, 1000, 0.8719), (0, 300, 0.575),
    (0, 100000, 0.1), (0, 100000, 0.9), (0, 4000000000, 1e-8), (0, 4294967295, 0.37), (0, 1, 0.3), (0, 2, 0.5),
    (0, 255, 0.5), (0, 256, 0.5), (0, 600, 0.21),
    (1, 1000, 0.0003), (1, 1000, 0.005), (1, 1000, 0.0159), (1, 1000, 0.0161), (1, 40, 0.39), (1, 40, 0.41),
    (1, 1000, 0.9841), (1, 1000, 0.9839), (1, 100, 0.5), (1, 33, 0.5), (1, 31, 0.5), (1, 1000000, 0.01),
    (1, 1000000, 0.5), (1, 4294967295, 0.002), (1, 3000000000, 0.9999999), (1, 20, 0.05), (1, 10000, 0.3),
]


def test_table_exponential_of_spec3():
    """orc_texp (the exponential behind f0 = exp(-n ln(1 + r)) of spec 3): < 1 ulp-ish of libm over the range the samplers use"""
    import math
    rng = np.random.default_rng(0)
    for y in np.concatenate((-rng.uniform(0, 200, 20000), -np.linspace(0, 1e-3, 200), [-0.0, -1e-300, -699.0])):
        assert abs(cbind.texp(y) - math.exp(y)) <= 4e-16 * math.exp(y)
    assert cbind.texp(0.0) == 1.0 and cbind.texp(-800.0) == 0.0


# every case under the default specification (2) and under its table exp / log variant (3, selectable)
@pytest.mark.parametrize("kind,n,p,spec", [c + (2,) for c in BINOM_CASES] + [c + (3,) for c in BINOM_CASES])
def test_binomial_samplers_have_the_exact_pmf(kind, n, p, spec):
    draws = cbind.binom_test(kind, n, p, 1.0 - p, 0xC0FFEE00 + kind, NV, spec=spec)
    assert draws.max() <= n
    pv = chi2_vs_binom(draws, n, p)
    assert pv > 1e-4, "chi-square p = %.3g" % pv
    m, v = n * p, n * p * (1 - p)
    assert abs(draws.mean() - m) < 5.0 * np.sqrt(v / NV) + 1e-12
    # unnormalised odds are what the callers pass: the same variates for any scale of (wa, wb)
    again = cbind.binom_test(kind, n, 8.0 * p, 8.0 * (1.0 - p), 0xC0FFEE00 + kind, 1000, spec=spec)
    assert np.array_equal(again, draws[:1000])


MULT4_CASES = [
    (1, [0.7, 0.1, 0.1, 0.1]), (3, [0.25, 0.25, 0.3, 0.2]), (50, [0.97, 0.01, 0.01, 0.01]), (50, [0.1, 0.2, 0.3, 0.4]),
    (200, [0.25, 0.25, 0.25, 0.25]),             # ~150 reads off the heaviest base: > XS -> two more binomials
    (170, [0.26, 0.25, 0.25, 0.24]),             # m straddles XS = 128
    (5000, [0.01, 0.97, 0.01, 0.01]),            # rarer outcome mean 150 -> BTRS, then m > XS
    (4000, [0.01, 0.01, 0.01, 0.97]),            # mean 120 -> inversion, m around XS
    (100000, [0.6, 0.3, 0.0999, 0.0001]), (1000, [0.5, 0.5, 0.0, 0.0]), (1000, [0.0, 0.0, 1.0, 0.0]),
    (300, [1e-9, 0.3, 0.3, 0.4]),
]


@pytest.mark.parametrize("x,W,spec", [c + (2,) for c in MULT4_CASES] + [c + (3,) for c in MULT4_CASES[1::2]])
def test_mult4_has_the_exact_multinomial_law(x, W, spec):
    W = np.array(W, dtype=np.float64)
    p = W / W.sum()
    draws = cbind.mult4_test(x, W, 0xABCD1234, NV, spec=spec).astype(np.int64)
    assert (draws.sum(axis=1) == x).all()
    for a in range(4):
        if p[a] == 0.0:
            assert (draws[:, a] == 0).all()
        elif p[a] == 1.0:
            assert (draws[:, a] == x).all()
        else:
            pv = chi2_vs_binom(draws[:, a], x, p[a])
            assert pv > 1e-4, (a, pv)
    # pairs: n_a + n_b ~ Binomial(x, p_a + p_b) -- wrong correlations between the categories would show here
    for a in range(4):
        for b in range(a + 1, 4):
            if 0.0 < p[a] + p[b] < 1.0:
                pv = chi2_vs_binom(draws[:, a] + draws[:, b], x, p[a] + p[b])
                assert pv > 1e-4, (a, b, pv)
    # the whole joint pmf when it is small enough to enumerate
    if x <= 3:
        code = draws @ np.array([(x + 1) ** 3, (x + 1) ** 2, x + 1, 1])
        pmf = np.zeros((x + 1) ** 4)
        for n0 in range(x + 1):
            for n1 in range(x + 1 - n0):
                for n2 in range(x + 1 - n0 - n1):
                    n3 = x - n0 - n1 - n2
                    pmf[n0 * (x + 1) ** 3 + n1 * (x + 1) ** 2 + n2 * (x + 1) + n3] = st.multinomial.pmf([n0, n1, n2, n3], x, p)
        assert chi2_vs_pmf(code, pmf) > 1e-4
    # scale of the weights does not matter
    assert np.array_equal(cbind.mult4_test(x, 3.5 * W, 0xABCD1234, 500, spec=spec), draws[:500].astype(np.uint32))


@pytest.mark.parametrize("G,S,depth,spec", [(3, 2, 60, 2), (8, 2, 2000, 2), (10, 2, 300, 2), (12, 2, 40000, 2), (5, 3, 3, 2),
                                             (8, 2, 2000, 3), (5, 3, 3, 3)])
def test_stage2_halving_tree_has_the_exact_law(G, S, depth, spec):
    """stage 2 (oracle/stats_agg.c: stage2_sample) from HAND-BUILT subset counts: with eta = I stage 1 is deterministic
    (every read's true base is its observed base), so N[s][H] is known exactly and
        sum_mu[s,g] = sum over the subsets H containing g of Binomial(N[s][H]; gamma[s,g] / Gamma_H)   (independent terms),
    whose pmf is a convolution.  Checked by chi-square over `nd` independent draws (the iteration counter keys the streams)."""
    rng = np.random.default_rng(100 + G)
    V = 24
    tau_idx = rng.integers(0, 4, size=(V, G)).astype(np.uint8)
    gamma = rng.dirichlet(np.full(G, 0.7), size=S)
    eta = np.eye(4)
    counts = np.zeros((V, S, 4), dtype=np.int64)
    for v in range(V):
        present = np.unique(tau_idx[v])
        for s in range(S):
            counts[v, s, present] = rng.poisson(depth, size=present.size)
    nd = 20000 if G <= 10 else 6000
    mus = np.empty((nd, S, G), dtype=np.int64)
    for it in range(nd):
        mu, E, nt = cbind.stats_agg(tau_idx, gamma, eta, counts, 99, it, want_ntab=True, spec=spec)
        mus[it] = mu
        if it == 0:
            nt0 = nt.copy()
            assert np.array_equal(E, np.diag(counts.sum(axis=(0, 1))).astype(np.uint64))
        else:
            assert np.array_equal(nt, nt0)                           # stage 1 is deterministic here
    # the hand-built table, from the definition
    N = np.zeros((S, 1 << G), dtype=np.int64)
    for v in range(V):
        for a in range(4):
            H = sum(1 << g for g in range(G) if tau_idx[v, g] == a)
            if H:
                N[:, H] += counts[v, :, a]
    assert np.array_equal(N, nt0.astype(np.int64))
    assert (mus.sum(axis=2) == counts.sum(axis=(0, 2))[None, :]).all()
    ps = []
    for s in range(S):
        Hs = np.nonzero(N[s])[0]
        for g in range(G):
            ns, pr = [], []
            for H in Hs:
                if H >> g & 1:
                    GamH = sum(gamma[s, j] for j in range(G) if H >> j & 1)
                    ns.append(N[s, H]); pr.append(gamma[s, g] / GamH)
            if not ns:
                assert (mus[:, s, g] == 0).all()
                continue
            mean = sum(n * p for n, p in zip(ns, pr))
            var = sum(n * p * (1 - p) for n, p in zip(ns, pr))
            d = mus[:, s, g]
            assert abs(d.mean() - mean) < 5.0 * np.sqrt(var / nd) + 1e-9
            if sum(ns) <= 60000:
                ps.append(chi2_vs_pmf(d, np.pad(sum_of_binomials_pmf(ns, pr), (0, 1)), min_expected=25.0))
            else:                                                    # normal regime: variance by chi-square of (n-1) s^2 / sigma^2
                q = (nd - 1) * d.var(ddof=1) / var
                ps.append(2.0 * min(st.chi2.cdf(q, nd - 1), st.chi2.sf(q, nd - 1)))
    ps = np.array(ps)
    assert ps.min() * ps.size > 1e-3, np.sort(ps)[:5]


@pytest.mark.parametrize("name", sorted(LAW_CASES))
@pytest.mark.parametrize("spec", [2, 3, 1, 4])
def test_specification_has_the_law_of_the_reference_sampleMu(name, spec):
    """orc_stats_agg (spec 3, spec 2; spec 4 = spec 2 over tau words) / orc_stats_counter (spec 1) against the reference's sampleMu,
    >= 2000 draws each"""
    if spec == 4 and LAW_CASES[name][2] > 8:
        pytest.skip("spec 4 is spec 2 itself above G = 8")
    counts, tau, gamma, eta = law_case(name)
    idx = cbind.onehot_to_idx(tau)
    n = 2000
    mu_r, E_r = reference_draws(name, n)
    mus, es = [], []
    for it in range(n):
        mu, E = (cbind.stats_agg(idx, gamma, eta, counts, 31415, it, spec=spec) if spec >= 2 else
                 cbind.stats_counter(idx, gamma, eta, counts, 31415, it))
        mus.append(mu.astype(np.int64)); es.append(E.astype(np.int64))
    assert_same_law(np.array(mus), np.array(es), mu_r, E_r, (name, spec))
    # and both against the exact conditional means
    e_mu, v_mu, e_E = cbind.stats_expect(idx, gamma, eta, counts)
    z = (np.mean(mus, axis=0) - e_mu) / np.sqrt(v_mu / n + 1e-12)
    assert np.abs(z).max() < 5.0
    np.testing.assert_allclose(np.mean(es, axis=0), e_E, rtol=0.02, atol=5.0 * np.sqrt(e_E.max() / n) + 0.5)
