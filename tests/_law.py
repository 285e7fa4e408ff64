"""Shared helpers of the law (T1) tests: chi-square of draws against an exact pmf, two-sample chi-square of two sets of
draws, and the cases on which the counter-based mu/E samplers are compared with the reference's own sampleMu
(/root/reference/desman/HaploSNP_Sampler.py:284-309, restated RandomState-exactly in oracle/ref_numpy.py: sample_mu and pinned
there by the golden fixtures of tests/test_oracle_golden.py)."""
import functools

import numpy as np
from scipy import stats as st


def chi2_vs_binom(draws, n, p, nbins=40, min_expected=50.0):
    """p-value of `draws` (integers) against Binomial(n, p): bins cut at the exact quantiles (so the test works for any n),
    merged until every bin expects >= min_expected draws."""
    draws = np.asarray(draws, dtype=np.int64)
    N = draws.size
    nbins = int(max(2, min(nbins, N / (2.0 * min_expected))))
    qs = np.unique(st.binom.ppf(np.linspace(0.0, 1.0, nbins + 1)[1:-1], n, p)).astype(np.int64)   # upper edges (inclusive)
    edges = np.concatenate(([-1], qs, [n]))
    edges = np.unique(edges)
    cdf = st.binom.cdf(edges, n, p)
    cdf[0], cdf[-1] = 0.0, 1.0
    exp = np.diff(cdf) * N
    obs = np.histogram(draws, bins=edges.astype(np.float64) + 0.5)[0].astype(np.float64)
    assert obs.sum() == N, "draws outside 0..n"
    # merge small bins from the left
    e2, o2, ea, oa = [], [], 0.0, 0.0
    for e, o in zip(exp, obs):
        ea += e; oa += o
        if ea >= min_expected:
            e2.append(ea); o2.append(oa); ea, oa = 0.0, 0.0
    if ea > 0.0:
        if e2:
            e2[-1] += ea; o2[-1] += oa
        else:
            e2.append(ea); o2.append(oa)
    e2, o2 = np.array(e2), np.array(o2)
    if e2.size < 2:
        return 1.0 if o2[0] == N else 0.0
    stat = ((o2 - e2) ** 2 / e2).sum()
    return float(st.chi2.sf(stat, e2.size - 1))


def chi2_vs_pmf(draws, pmf, min_expected=50.0):
    """p-value of integer `draws` against the explicit pmf over 0..len(pmf)-1 (bins merged to >= min_expected)."""
    draws = np.asarray(draws, dtype=np.int64)
    N = draws.size
    assert draws.min() >= 0 and draws.max() < len(pmf)
    obs = np.bincount(draws, minlength=len(pmf)).astype(np.float64)
    exp = np.asarray(pmf, dtype=np.float64) * N
    e2, o2, ea, oa = [], [], 0.0, 0.0
    for e, o in zip(exp, obs):
        ea += e; oa += o
        if ea >= min_expected:
            e2.append(ea); o2.append(oa); ea, oa = 0.0, 0.0
    if e2:
        e2[-1] += ea; o2[-1] += oa
    else:
        e2.append(ea); o2.append(oa)
    e2, o2 = np.array(e2), np.array(o2)
    e2 *= N / e2.sum()                      # a truncated pmf tail
    if e2.size < 2:
        return 1.0
    return float(st.chi2.sf(((o2 - e2) ** 2 / e2).sum(), e2.size - 1))


def chi2_two_sample(a, b, nbins=10):
    """p-value of the hypothesis that integer samples a and b come from one distribution (pooled-quantile bins)."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    pooled = np.concatenate((a, b))
    edges = np.unique(np.quantile(pooled, np.linspace(0, 1, nbins + 1)[1:-1]))
    ca = np.bincount(np.searchsorted(edges, a, side="left"), minlength=edges.size + 1).astype(np.float64)
    cb = np.bincount(np.searchsorted(edges, b, side="left"), minlength=edges.size + 1).astype(np.float64)
    keep = (ca + cb) > 0
    ca, cb = ca[keep], cb[keep]
    if ca.size < 2:
        return 1.0
    na, nb = ca.sum(), cb.sum()
    ea, eb = (ca + cb) * na / (na + nb), (ca + cb) * nb / (na + nb)
    stat = ((ca - ea) ** 2 / ea).sum() + ((cb - eb) ** 2 / eb).sum()
    return float(st.chi2.sf(stat, ca.size - 1))


def sum_of_binomials_pmf(ns, ps):
    """exact pmf of a sum of independent Binomial(n_i, p_i)"""
    pmf = np.array([1.0])
    for n, p in zip(ns, ps):
        if n == 0 or p <= 0.0:
            continue
        pmf = np.convolve(pmf, st.binom.pmf(np.arange(int(n) + 1), int(n), float(p)))
    return pmf


# ---- cases for "spec = reference in law" ------------------------------------------------------------------------------
# name -> (V, S, G, depth_scale, kind of state)
LAW_CASES = {
    "G3": (30, 6, 3, 1.0, "random"),                    # the shape the first law test used
    "G10": (16, 3, 10, 1.0, "random"),                  # stage 2 as its own launch (G >= 10)
    "G12": (12, 2, 12, 1.0, "random"),
    "deep": (20, 4, 4, 15.0, "random"),                 # x 15 depth: BTRS in stage 1, deferred lists, stats_big_kernel
    "converged": (40, 4, 4, 4.0, "truth"),              # eta ~ 0.97 I, tau = the generating haplotypes: rare-outcome inversion
    "words": (80, 4, 2, 3.0, "truth"),                  # 80 positions on at most 16 tau words: what spec 4 pools (5 positions a word)
}


def law_case(name):
    """(counts, tau one-hot, gamma, eta) of a case"""
    from desman_amd.synth import synth_counts, random_state
    from oracle import cbind
    V, S, G, depth, kind = LAW_CASES[name]
    seed = 4000 + sorted(LAW_CASES).index(name)
    counts, tau_true, gamma_true = synth_counts(V, S, G, seed=seed, depth_scale=depth)
    if kind == "truth":
        tau = cbind.idx_to_onehot(tau_true)
        gamma = np.ascontiguousarray(gamma_true)
        eta = 0.96 * np.eye(4) + 0.01
    else:
        tau, gamma, eta = random_state(V, S, G, seed=seed + 1)
    return counts, tau, gamma, eta


@functools.lru_cache(maxsize=None)
def reference_draws(name, n):
    """n draws of (sum_mu [S,G], Esum [4,4] = [observed][true]) by the reference's sampleMu"""
    from oracle import ref_numpy as rn
    counts, tau, gamma, eta = law_case(name)
    rs = np.random.RandomState(777)
    mus, es = [], []
    for _ in range(n):
        E, mu = rn.sample_mu(rs, tau, gamma, eta, counts)
        mus.append(mu.sum(axis=(0, 2)))
        es.append(E.sum(axis=(0, 1)))
    return np.array(mus), np.array(es)


def assert_same_law(mu_a, E_a, mu_b, E_b, what):
    """per-(s,g) marginals of sum_mu and the 16 entries of Esum: two-sample chi-square (Bonferroni over all of them),
    means by z-test, variances within Monte-Carlo error"""
    mu_a, mu_b = np.asarray(mu_a, dtype=np.float64), np.asarray(mu_b, dtype=np.float64)
    E_a, E_b = np.asarray(E_a, dtype=np.float64), np.asarray(E_b, dtype=np.float64)
    cols_a = np.concatenate((mu_a.reshape(len(mu_a), -1), E_a.reshape(len(E_a), -1)), axis=1)
    cols_b = np.concatenate((mu_b.reshape(len(mu_b), -1), E_b.reshape(len(E_b), -1)), axis=1)
    ps = []
    for j in range(cols_a.shape[1]):
        a, b = cols_a[:, j], cols_b[:, j]
        if a.std() == 0.0 and b.std() == 0.0:
            assert a[0] == b[0], (what, j)
            continue
        ps.append(chi2_two_sample(a, b))
        se = np.sqrt(a.var() / a.size + b.var() / b.size)
        assert abs(a.mean() - b.mean()) < 5.0 * se + 1e-9, (what, j, a.mean(), b.mean(), se)
        va, vb = a.var(ddof=1), b.var(ddof=1)
        # var of a sample variance ~ 2 sigma^4 / (n - 1) for near-normal sums (kurtosis-padded by the factor 2)
        tol = 5.0 * np.sqrt(2.0 * 2.0 * (max(va, vb) ** 2) * (1.0 / (a.size - 1) + 1.0 / (b.size - 1)))
        assert abs(va - vb) < tol + 1e-9, (what, j, va, vb, tol)
    ps = np.array(ps)
    assert ps.min() * ps.size > 1e-3, (what, "chi-square: smallest of %d p-values %.3g" % (ps.size, ps.min()))
    assert (ps < 0.01).mean() < 0.08, (what, "too many small p-values", np.sort(ps)[:8])
