"""Seeded subsets of the fuzzers (scripts/dbg/fuzz_gibbs.py, fuzz_nmft.py) and the cross-process determinism soak
(scripts/dbg/soak_determinism.py) inside the driver-run suite: the boundaries of every kernel selection first, then random shapes.

* Gibbs: the WHOLE device iteration against the oracle composition in the reference's order (HaploSNP_Sampler.py:341-358; tau sweep
  c_sample_tau.c:95-204 bit for bit) -- lane-group widths S = 16/17, 32/33, 48, 64/65, 96/97, 128/129, 192/193, 256/257, 384/385, the
  stage-2 switch G = 9/10, the aggregated specs' end G = 16/17, the single-haplotype and single-sample cases.
* NMFT: factors, update count, objective trace and get_tau against the C oracle (Init_NMFT.py:98-245) at the tile boundaries of the
  matrix-core kernels (S = 16 k, 128/129, 256/257, 512; G = 4/5, 8/9, 12/13, 16; V not a multiple of four).
* Determinism: three fresh processes -- two with the run-time placement / ordering heuristics on, one with them off -- end on the
  same bits: where and when things run differs from process to process, never what is computed."""
import os
import subprocess
import sys

import numpy as np
import pytest

from desman_amd import _lib
from desman_amd.synth import synth_counts
from oracle import cbind, ref_numpy as rn

import test_gpu_parity as tp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gibbs_shapes():
    rs = np.random.RandomState(11)
    edge = [(150, S, 4) for S in (1, 16, 17, 33, 48, 65, 97, 129, 193, 257, 385)] + [(260, 24, G) for G in (1, 9, 10, 16, 17)] + \
           [(V, 64, 8) for V in (1, 3, 257)]
    rnd = [(int(rs.randint(1, 700)), int(rs.randint(1, 140)), int(rs.randint(1, 13))) for _ in range(6)]
    return edge + rnd


@pytest.mark.parametrize("V,S,G", _gibbs_shapes())
def test_fuzz_whole_gibbs_iteration_against_the_oracle(V, S, G):
    for spec in (2, 1) if (V + S + G) % 3 else (3, 1):          # every third shape takes the table exp / log variant of the aggregated spec
        if G > 16 and spec >= 2:
            continue                                            # the aggregated specifications stop at 16 haplotypes
        ctx = _lib.Context(0)
        try:
            tp.test_gibbs_update_is_self_consistent_with_oracle(ctx, V, S, G, 3, spec)
        finally:
            ctx.close()


def _nmft_shapes():
    rs = np.random.RandomState(7)
    edge = [(203, S, 3) for S in (1, 16, 17, 65, 97, 128, 129, 257, 512)] + [(77, S, 13) for S in (15, 64, 112, 193, 289)] + \
           [(501, 200, G) for G in (1, 5, 9, 16)] + [(V, 300, 7) for V in (1, 3, 5, 1025)]
    rnd = [(int(rs.randint(1, 3000)), int(rs.randint(1, 513)), int(rs.randint(1, 17))) for _ in range(6)]
    return edge + rnd


@pytest.mark.parametrize("V,S,G", _nmft_shapes())
def test_fuzz_nmft_against_the_oracle(V, S, G):
    counts, _, _ = synth_counts(V, S, min(G, 4), seed=V + S)
    tau0, gam0 = rn.nmft_random_initialize(np.random.RandomState(V + 3 * S + G), V, S, G)
    F = cbind.nmft_freq(counts)
    for fix_gamma in (False, True):
        tc, gc = tau0.copy(), gam0.copy()
        n_ref, tr_ref = (cbind.nmft_factorize_tau if fix_gamma else cbind.nmft_factorize)(F, tc, gc, max_iter=12, min_change=1e-5)
        if np.isnan(tc).any() or np.isnan(tr_ref[:n_ref]).any():
            continue          # a start with exact zeros in all four bases of a haplotype: 0/0 in the reference as well
        c = _lib.Context(0)
        c.set_counts(counts)
        c.nmft_set(tau0, gam0)
        n, tr = c.nmft_factorize(max_iter=12, min_change=1e-5, fix_gamma=fix_gamma)
        t, g = c.nmft_get()
        oh = c.nmft_get_tau()
        c.close()
        assert n == n_ref
        np.testing.assert_allclose(tr[:n], tr_ref[:n_ref], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(t, tc, rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(g, gc, rtol=1e-6, atol=1e-12)
        assert np.array_equal(oh, cbind.idx_to_onehot(cbind.nmft_get_tau(np.ascontiguousarray(t), G)))


_SOAK = r'''
import sys, hashlib; sys.path.insert(0, %r)
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
V, S, G = (int(x) for x in sys.argv[1:4])
counts, _, _ = synth_counts(V, S, G, 1234)
ctx = _lib.Context(0); ctx.set_counts(counts); ctx.seed(5)
rs = np.random.RandomState(0)
gam0 = np.ascontiguousarray(rs.dirichlet(np.full(G, 0.01), size=S).T)
d = rs.dirichlet(np.full(4, 0.01), size=V * G).reshape(V, G, 4)
tau0 = np.ascontiguousarray(np.transpose(d, (2, 0, 1)).reshape(4 * V, G))
ctx.nmft_set(tau0, gam0); ctx.nmft_factorize(max_iter=300, min_change=0.0)
t = ctx.nmft_get_tau(); _, g = ctx.nmft_get()
ctx.set_state(t, np.ascontiguousarray(g.T), 0.96 * np.eye(4) + 0.01)
h = hashlib.sha256()
for n in (7, 300, 1, 193, 20, 479):
    ctx.gibbs_update(n)
    tr = ctx.get_trace()
    for k in ("ll", "lp", "nchange", "gamma", "eta"): h.update(np.ascontiguousarray(tr[k]).tobytes())
    tt, gg, ee = ctx.get_state(); h.update(tt.tobytes()); h.update(gg.tobytes()); h.update(ee.tobytes())
print(h.hexdigest(), tr["ll"][-1])
''' % ROOT


@pytest.mark.parametrize("V,S,G", [(10000, 64, 8), (6000, 96, 5)])
def test_three_processes_end_on_the_same_bits(V, S, G):
    """1000 iterations in calls of uneven length after a 300-update NMF start (the benchmark's chain; the second shape takes the
    register-lean sweep and a table whose rows are not whole 256 B blocks).  DESMAN_HIP_NTAB_TUNE=0 / DESMAN_HIP_TAU_ORDER=0 switch the
    measured table place and the fp64-blocks-first order off: the bits may not depend on either."""
    outs = []
    for env in ({}, {"DESMAN_HIP_NTAB_TUNE": "0", "DESMAN_HIP_TAU_ORDER": "0"}, {}):
        r = subprocess.run([sys.executable, "-c", _SOAK, str(V), str(S), str(G)], env=dict(os.environ, **env), capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.split()[0])
    assert len(set(outs)) == 1 and len(outs[0]) == 64, outs


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 6: operands at the ends of the exponent range and degenerate inputs for every kernel that divides or takes a
# logarithm.  The NaN rows of rounds 2-4 (factorize_tau from subnormal start values) were found by a script OUTSIDE the suite; these legs
# are inside it.  Each compares with the oracle, NaN-aware, exact where the result is an integer.
# ---------------------------------------------------------------------------------------------------------------------------------
def _extreme_pi(rs, S, G, kind):
    pi = rs.dirichlet(np.ones(G), size=S)
    if kind == "masked":                     # Eta_Sampler.py:355-369: gamma masked by the gene's presence, exact zeros, NOT renormalised
        pi[:, rs.rand(G) < 0.5] = 0.0
    elif kind == "tiny":
        pi[rs.rand(S, G) < 0.3] = 1e-300
    elif kind == "subnormal":
        m = rs.rand(S, G) < 0.3
        pi[m] = rs.choice([5e-324, 1e-310, 2.2e-308], size=int(m.sum()))
    elif kind == "zero_sample":              # a sample nobody is in: every mixture of that sample is zero
        pi[rs.randint(S), :] = 0.0
        pi[rs.rand(S, G) < 0.2] = 0.0
    elif kind == "one_hot":                  # abundances that are exactly 0 or 1
        pi = np.zeros((S, G)); pi[np.arange(S), rs.randint(G, size=S)] = 1.0
    return np.ascontiguousarray(pi)


def _extreme_eta(rs, kind):
    if kind == "identity":                   # exact zeros off the diagonal: a base no haplotype carries has mixture zero
        return np.eye(4)
    if kind == "zero_row":
        e = 0.96 * np.eye(4) + 0.01; e[2, :] = 0.0; return e
    if kind == "tiny":
        e = np.full((4, 4), 1e-300); e[np.arange(4), np.arange(4)] = 1.0; return e
    if kind == "subnormal":
        e = np.full((4, 4), 5e-324); e[np.arange(4), np.arange(4)] = 1.0; return e
    return 0.96 * np.eye(4) + 0.01


@pytest.mark.parametrize("pi_kind,eta_kind", [("masked", "usual"), ("masked", "identity"), ("tiny", "usual"), ("subnormal", "usual"), ("zero_sample", "usual"),
                                              ("one_hot", "identity"), ("usual", "zero_row"), ("usual", "tiny"), ("subnormal", "subnormal"), ("tiny", "tiny")])
@pytest.mark.parametrize("V,S,G", [(61, 9, 3), (40, 64, 8), (33, 33, 5)])
def test_fuzz_tau_sweep_through_the_shim_with_degenerate_abundances(V, S, G, pi_kind, eta_kind):
    """c_sample_tau.c:136-169 is plain IEEE: log(0) = -inf, 0 * -inf = NaN, and both propagate into the candidate sums, the softmax and the
    inverse-CDF walk (whose last edge is forced).  The shim takes the caller's pi / eta verbatim (Eta_Sampler.py:147-157, 355-369 passes a
    masked gamma with exact zeros; nothing clamps at 1e-6 there), so the device sweep must walk the same way: tau and the flip count of
    three consecutive sweeps are the oracle's."""
    import desman_amd.sampletau as st
    rs = np.random.RandomState(V * 1000 + S * 10 + G + len(pi_kind) * 7 + len(eta_kind))
    counts, _, _ = synth_counts(V, S, min(G, 4), seed=V + S)
    counts[rs.rand(V, S) < 0.1] = 0                                   # cells without a read: 0 * log(.) terms
    counts = np.ascontiguousarray(counts)
    tau0 = cbind.idx_to_onehot(rs.randint(4, size=(V, G)).astype(np.uint8))
    pi = _extreme_pi(rs, S, G, pi_kind)
    eta = _extreme_eta(rs, eta_kind)
    seed = 4242 + V
    u = cbind.MT19937(seed).uniform(3 * V * G)
    ref = tau0.copy()
    n_ref = [cbind.sample_tau_u(ref, pi, eta, counts, u[i * V * G:(i + 1) * V * G]) for i in range(3)]
    got = tau0.copy()
    st.initRNG(); st.setRNG(seed)
    with np.errstate(all="ignore"):
        n = [st.sample_tau(got, pi, eta, counts) for _ in range(3)]
    st.freeRNG()
    assert n == n_ref, (n, n_ref)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("case", ["zeros", "single_base", "huge", "ties", "p_edges", "eta_zero", "mixed"])
def test_fuzz_lrt_kernel_with_degenerate_frequencies(case):
    """Variant_Filter.py:320-390 per position: a bounded 1-D minimisation of a two-component mixture NLL.  Positions with no read at all,
    with every read on one base (the second base's count is 0), with 1e8 reads, with the two largest counts equal, start values on the
    bounds, and an error matrix with exact zeros (log(0) inside the NLL): the kernel's (p, NLL of the mixture, NLL of the null) against
    the restated step (oracle/ref_numpy.py: lrt_step, scipy's bounded Brent restated), NaN / inf in the same places."""
    rs = np.random.RandomState(len(case))
    V = 48
    freq = rs.poisson(40, size=(V, 4)).astype(np.int64)
    eta = 0.96 * np.eye(4) + 0.01
    if case == "zeros":
        freq[::3] = 0
    elif case == "single_base":
        freq[:] = 0; freq[np.arange(V), rs.randint(4, size=V)] = rs.randint(1, 500, size=V)
    elif case == "huge":
        freq *= 2500000
    elif case == "ties":
        freq[:, 1] = freq[:, 0]; freq[:, 2:] //= 8
    elif case == "eta_zero":
        eta = np.eye(4)
    elif case == "mixed":
        freq[::4] = 0; freq[1::4, :] = [7, 0, 0, 0]; freq[2::4, :] = [10 ** 8, 10 ** 8, 1, 0]
    maxA = np.argmax(freq, axis=1)
    ft = freq.copy(); ft[np.arange(V), maxA] = -1
    maxB = np.argmax(ft, axis=1)
    ff = freq.astype(np.float64)
    p0 = np.minimum(freq.max(axis=1) / np.maximum(freq.sum(axis=1), 1), 0.99)
    if case == "p_edges":
        p0[::2] = 0.99; p0[1::2] = 0.5
    with np.errstate(all="ignore"):
        pg, mg, bg = _lib.lrt_step(ff, maxA, maxB, eta, 0.99, True, p0)
        po, mo, bo = rn.lrt_step(ff, maxA, maxB, eta, 0.99, True, p0)
    for name, g, o, tol in (("p", pg, po, dict(rtol=0, atol=1e-9)), ("mixture NLL", mg, mo, dict(rtol=1e-12, atol=1e-9)), ("null NLL", bg, bo, dict(rtol=1e-12, atol=1e-9))):
        assert np.array_equal(np.isnan(g), np.isnan(o)), (case, name, np.where(np.isnan(g) != np.isnan(o)))
        fin = np.isfinite(o)
        assert np.array_equal(np.isfinite(g), fin), (case, name, np.where(np.isfinite(g) != fin))
        assert np.array_equal(g[~fin & ~np.isnan(o)], o[~fin & ~np.isnan(o)]), (case, name)          # the same infinities
        np.testing.assert_allclose(g[fin], o[fin], err_msg="%s %s" % (case, name), **tol)


@pytest.mark.parametrize("case", ["eta_exact_01", "delta_zero_column", "tiny", "cov_huge", "nobody"])
def test_fuzz_kl_assign_with_degenerate_operands(case):
    """GeneAssign's KL assignment (multiplicative updates of eta under cov ~ eta . delta^T; oracle/ref_genes.py: kl_assign): start values
    that are exactly 0 and 1, a haplotype nobody carries (a zero column of delta), abundances of 1e-308, coverages of 1e12, genes without
    any coverage -- the kernel's eta, update count and divergence against the restated loop, NaN in the same places."""
    from oracle import ref_genes as rg
    rng = np.random.default_rng(len(case) + 5)
    C, S, G = 41, 13, 4
    delta = rng.random((S, G)) * 50.0
    truth = (rng.random((C, G)) < 0.5).astype(float)
    cov = rng.poisson(truth @ delta.T + 0.3).astype(float)
    eta0 = rng.random((C, G))
    if case == "eta_exact_01":
        eta0 = (rng.random((C, G)) < 0.5).astype(float)
    elif case == "delta_zero_column":
        delta[:, 2] = 0.0
    elif case == "tiny":
        delta[:, 1] = 1e-308; eta0[::3, :] = 1e-308
    elif case == "cov_huge":
        cov *= 1e12
    elif case == "nobody":
        cov[::2, :] = 0.0; delta[0, :] = 0.0
    with np.errstate(all="ignore"):
        eta, n, div = _lib.kl_assign(cov, delta, eta0, max_iter=300)
        ref_eta, ref_n, ref_div = rg.kl_assign(cov, delta, eta0, max_iter=300)
    assert np.array_equal(np.isnan(eta), np.isnan(ref_eta)), (case, int(np.isnan(eta).sum()), int(np.isnan(ref_eta).sum()))
    assert np.isnan(div) == np.isnan(ref_div), (case, div, ref_div)
    if not np.isnan(ref_div):
        assert abs(n - ref_n) <= 2, (case, n, ref_n)
        assert abs(div - ref_div) <= 1e-6 * max(1.0, abs(ref_div)), (case, div, ref_div)
    fin = ~np.isnan(ref_eta)
    np.testing.assert_allclose(eta[fin], ref_eta[fin], rtol=1e-4, atol=1e-6, err_msg=case)


@pytest.mark.parametrize("S,G", [(5, 3), (64, 8), (33, 12)])
def test_fuzz_dirichlet_draws_with_empty_rows_and_counts_near_2_to_31(S, G):
    """HaploSNP_Sampler.py:263-281 with sums at both ends: samples whose mu sums are all zero (every shape is the prior's 0.1: the boost,
    the clamp at 1e-6 and the renormalisation decide the row), a single 1 among zeros, counts of 2^31 - 1 and 2^40 next to zeros (the
    normalised variates of the small shapes underflow towards the clamp), an all-zero Esum -- value by value against the draw
    specification (oracle: orc_dirichlet_counter)."""
    ctx = _lib.Context(0)
    try:
        V = 4
        counts, _, _ = synth_counts(V, S, max(G, 2), seed=52)
        tau = cbind.idx_to_onehot(np.zeros((V, G), np.uint8))
        ctx.set_counts(counts)
        ctx.set_state(tau, np.full((S, G), 1.0 / G), 0.96 * np.eye(4) + 0.01)
        seed = 0xA5A5A5A5DEADBEEF
        ctx.seed(1, ctr_seed=seed)
        big = np.uint64(2 ** 31 - 1)
        sum_mu = np.zeros((S, G), np.uint64)
        sum_mu[1 % S, 0] = 1
        sum_mu[2 % S, :] = big
        sum_mu[3 % S, ::2] = big
        sum_mu[4 % S, G - 1] = np.uint64(2 ** 40)
        for esum in (np.zeros((4, 4), np.uint64), np.diag([big] * 4).astype(np.uint64), np.full((4, 4), big, np.uint64),
                     np.array([[0, 1, 0, 2 ** 40], [0, 0, 0, 0], [big, 0, 0, 0], [1, 1, 1, 1]], np.uint64)):
            for it in (0, 77):
                g, e = ctx.draw_gamma_eta(it, sum_mu, esum)
                g_ref, e_ref, _ = cbind.dirichlet_counter(sum_mu, esum, seed, it)
                assert np.isfinite(g).all() and np.isfinite(e).all()
                np.testing.assert_allclose(g, g_ref, rtol=1e-13, atol=0)
                np.testing.assert_allclose(e, e_ref, rtol=1e-13, atol=1e-300)
    finally:
        ctx.close()


@pytest.mark.parametrize("V,S,G", [(515, 96, 2), (303, 64, 4), (640, 48, 3), (700, 128, 8), (200, 300, 4), (97, 20, 12)])
def test_fuzz_factorize_with_subnormal_start_values_gamma_updating(V, S, G):
    """Init_NMFT.py:98-115 (`factorize`: gamma updating, `_adjustment` after every update) from the start values that made NaN rows of
    `factorize_tau` in rounds 2-4 (tests/test_gpu_parity.py: test_factorize_tau_with_subnormal_start_values_and_tiny_abundances): tau start
    values of 1e-308 / 1e-130, abundance columns of 1 / 1e-200, and a start whose gamma column is subnormal throughout."""
    counts, _, _ = synth_counts(V, S, min(G, 4), seed=V + S)
    tau0, gam0 = rn.nmft_random_initialize(np.random.RandomState(V + 3 * S + G), V, S, G)
    gam0 = np.full((G, S), 1.0 / G)
    for s in (3, S - 1):
        gam0[:, s] = 1e-200
        gam0[s % G, s] = 1.0
    gam0[:, 5 % S] = 1e-310
    for v in range(0, V, 7):
        tau0[3 * V + v, :] = 1e-130
        tau0[3 * V + v, v % G] = 1e-308
    F = cbind.nmft_freq(counts)
    ctx = _lib.Context(0)
    try:
        ctx.set_counts(counts)
        ctx.nmft_set(tau0, gam0)
        tc, gc = tau0.copy(), gam0.copy()
        with np.errstate(all="ignore"):
            n_ref, tr_ref = cbind.nmft_factorize(F, tc, gc, max_iter=6, min_change=0.0)
        n, tr = ctx.nmft_factorize(max_iter=6, min_change=0.0, fix_gamma=False)
        t, g = ctx.nmft_get()
        assert n == n_ref
        assert np.array_equal(np.isnan(t), np.isnan(tc)) and np.array_equal(np.isnan(g), np.isnan(gc))
        assert np.array_equal(np.isnan(tr[: n + 1]), np.isnan(tr_ref[: n_ref + 1]))
        ok = ~np.isnan(tr_ref[: n_ref + 1])
        np.testing.assert_allclose(tr[: n + 1][ok], tr_ref[: n_ref + 1][ok], rtol=1e-9)
        ft, fg = ~np.isnan(tc), ~np.isnan(gc)
        np.testing.assert_allclose(t[ft], tc[ft], rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(g[fg], gc[fg], rtol=1e-6, atol=1e-12)
    finally:
        ctx.close()


def _extreme_case(i, V, S, G, pi_kind, eta_kind):
    """the generator of scripts/dbg/fuzz_extreme.py (case i): counts of three depths with empty cells, a random tau, degenerate pi / eta"""
    rs = np.random.RandomState(i)
    counts, _, _ = synth_counts(V, S, min(G, 4), seed=i, depth_scale=float(rs.choice([0.05, 1.0, 30.0])))
    counts[rs.rand(V, S) < 0.1] = 0
    counts = np.ascontiguousarray(counts)
    tau0 = cbind.idx_to_onehot(rs.randint(4, size=(V, G)).astype(np.uint8))
    return counts, tau0, _extreme_pi(rs, S, G, pi_kind), _extreme_eta(rs, eta_kind)


@pytest.mark.parametrize("i,V,S,G,pi_kind,eta_kind", [(3, 202, 12, 14, "subnormal", "identity"), (71, 261, 23, 13, "masked", "identity"),
                                                      (194, 212, 5, 12, "masked", "identity"), (205, 267, 4, 3, "masked", "zero_row"),
                                                      (257, 20, 41, 10, "subnormal", "identity")])
def test_fuzz_padded_lanes_add_nothing_when_eta_has_exact_zeros(i, V, S, G, pi_kind, eta_kind):
    """Found by scripts/dbg/fuzz_extreme.py in round 6 (5 of 300 cases; had shipped since round 1): the lanes of a group beyond the last
    sample carry count 0 and abundance 1, so their mixture value is a sum of eta entries -- positive, hence 0 * log(.) = 0, for any error
    matrix WITHOUT exact zeros.  With the identity or a zero row (the shim takes the caller's eta verbatim) it is 0 and 0 * log(0) = NaN
    poisoned a candidate's total where c_sample_tau.c:152-169 has -inf or a finite number: other draws than the reference's.  The libm
    branch of the candidate sums and of the likelihood now skips those lanes (dsm_device.h: sweep_candidate).  Sweep, its
    log-probabilities and the log-likelihood of the state it leaves, against the oracle, NaN and -inf in the same places."""
    counts, tau0, pi, eta = _extreme_case(i, V, S, G, pi_kind, eta_kind)
    u = cbind.MT19937(i).uniform(V * G)
    ref = tau0.copy()
    with np.errstate(all="ignore"):
        n_ref, lp_ref = cbind.sample_tau_u(ref, pi, eta, counts, u, want_logp=True)
        ll_ref = cbind.loglik(cbind.onehot_to_idx(ref), pi, eta, counts)
    for screen in (True, False):
        ctx = _lib.Context(0)
        try:
            ctx.set_counts(counts); ctx.set_state(tau0, pi, eta)
            ctx.set_tau_rng(_lib.RNG_MT19937); ctx.set_mt_state(_lib.mt_seed_state(i))
            ctx.set_tau_screen(screen)
            n, lp = ctx.sample_tau(want_logp=True)
            got = ctx.get_state()[0]
            ll = ctx.loglik()[0]
        finally:
            ctx.close()
        assert n == n_ref and np.array_equal(got, ref)
        assert np.array_equal(np.isnan(lp), np.isnan(lp_ref))
        inf = np.isinf(lp_ref)
        assert np.array_equal(np.isinf(lp), inf) and np.array_equal(lp[inf], lp_ref[inf])
        fin = np.isfinite(lp_ref)
        np.testing.assert_allclose(lp[fin], lp_ref[fin], rtol=1e-12)
        assert (np.isnan(ll) and np.isnan(ll_ref)) or ll == ll_ref or abs(ll - ll_ref) <= 1e-12 * abs(ll_ref), (ll, ll_ref)


@pytest.mark.parametrize("pi_kind,eta_kind", [("tiny", "usual"), ("subnormal", "usual"), ("masked", "identity"), ("usual", "tiny"), ("subnormal", "subnormal"),
                                              ("tiny", "tiny"), ("one_hot", "zero_row"), ("zero_sample", "usual")])
@pytest.mark.parametrize("V,S,G", [(300, 64, 8), (211, 33, 5), (150, 96, 12)])
def test_fuzz_stats_pass_with_weights_at_the_ends_of_the_exponent_range(V, S, G, pi_kind, eta_kind):
    """Stage 1 of the mu/E pass divides three times per item (dsm_binom.h: s1_item).  Round 6 runs those divisions WITHOUT the scaling /
    fix-up instructions of the IEEE expansion while every lane's weights are within [2^-400, 2^400], and with them otherwise (one
    wave-uniform test per item): weights of 1e-300 and below, subnormal products, exact zeros (the fall-back to the abundances, a base
    nobody carries) must leave the sums the twin's (oracle/stats_agg.c divides with `/`), bit for bit, under every specification."""
    rs = np.random.RandomState(V + S + G + len(pi_kind) + 3 * len(eta_kind))
    counts, _, _ = synth_counts(V, S, min(G, 4), seed=V + S)
    counts = np.ascontiguousarray(counts)
    idx = rs.randint(4, size=(V, G)).astype(np.uint8)
    gamma = _extreme_pi(rs, S, G, pi_kind)
    eta = _extreme_eta(rs, eta_kind)
    ctx = _lib.Context(0)
    try:
        ctx.set_counts(counts); ctx.set_state(cbind.idx_to_onehot(idx), gamma, eta)
        seed = 0x0123456789ABCDEF
        ctx.seed(1, ctr_seed=seed)
        for spec in (2, 4, 3):
            ctx.force_stats_spec(spec)
            for it in (0, 5):
                mu, E = ctx.sample_stats(it)
                mu_ref, E_ref = cbind.stats_agg(idx, gamma, eta, counts, seed, it, spec=spec)
                assert np.array_equal(E, E_ref), (spec, it)
                assert np.array_equal(mu, mu_ref), (spec, it)
            ctx.force_stats_spec(0)
    finally:
        ctx.close()


def test_fuzz_nmft_divisions_at_both_ends_of_the_exponent_range():
    """kernels_nmft.hip: fdiv_ext brings its operands to within 2^+-512 of one by exact powers of two and puts the powers back on the quotient.
    Put back one after the other (round 5), a quotient of two HUGE or two TINY operands passed through the subnormal range or through
    infinity on the way although a / b itself is an ordinary number (tn = 2^501 over tot = 2^1023 in the row normalisation of the
    fixed-gamma update, Init_NMFT.py:180-181, which has no eps clamp).  Against IEEE division: within 1 ulp (fdiv is faithful, not always
    correctly rounded) wherever the quotient is a normal number, the same zero / infinity where it is not, over every pairing of ends."""
    rs = np.random.RandomState(5)
    ends = [2.0 ** e for e in (-1074, -1060, -1022, -900, -600, -501, -499, -300, -1, 0, 1, 300, 499, 501, 600, 900, 1022, 1023)]
    a = np.array([x * (1.0 + rs.rand()) for x in ends for _ in ends])
    b = np.array([y * (1.0 + rs.rand()) for _ in ends for y in ends])
    a = np.minimum(a, np.finfo(np.float64).max)
    b = np.minimum(b, np.finfo(np.float64).max)
    with np.errstate(all="ignore"):
        ref = a / b
    got = _lib.debug_fdiv(0, a, b)
    normal = np.isfinite(ref) & (np.abs(ref) >= 2.0 ** -1022)
    ulp = np.spacing(np.abs(ref[normal]))
    assert np.all(np.abs(got[normal] - ref[normal]) <= ulp), np.argwhere(np.abs(got[normal] - ref[normal]) > ulp)[:5]
    assert np.array_equal(np.isinf(got), np.isinf(ref))
    sub = np.isfinite(ref) & ~normal                             # subnormal or zero quotients: within one spacing of the subnormal grid
    assert np.all(np.abs(got[sub] - ref[sub]) <= 2.0 ** -1074 * 2)
    # the case of the advice, exactly: two huge operands, an ordinary quotient
    q = _lib.debug_fdiv(0, np.array([2.0 ** 501, 2.0 ** -600 * 3]), np.array([2.0 ** 1023, 2.0 ** -1000 * 7]))
    np.testing.assert_allclose(q, [2.0 ** -522, (2.0 ** 400) * 3 / 7], rtol=3e-16)
