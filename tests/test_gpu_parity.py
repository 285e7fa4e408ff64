"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and
the golden fixtures.  Integer outcomes (tau, nchange, auxiliary-count sums) must
be bit-exact; floating point within the tolerance written in each test.
Run on an MI355X with:  python -m pytest tests -m gpu
"""
import glob
import os

import numpy as np
import pytest

from desman_amd import _lib
from desman_amd.synth import synth_counts, random_state
from oracle import cbind, ref_numpy as rn

from _law import LAW_CASES, assert_same_law, law_case, reference_draws

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


@pytest.fixture(params=[2, 3, 1, 4], ids=["aggregated-muE", "aggregated-muE-v3", "per-read-muE", "aggregated-over-words"])
def spec_ctx(ctx, request):
    """the shared context with one of the two mu/E specifications forced (small test shapes would all take the
    per-read pass by the shape rule): the law / chain-level tests must hold for both"""
    ctx.force_stats_spec(request.param)
    yield ctx
    ctx.force_stats_spec(0)


def _load(ctx, counts, tau, gamma, eta, mt_seed=None):
    ctx.set_counts(counts)
    ctx.set_state(tau, gamma, eta)
    if mt_seed is not None:
        ctx.seed(mt_seed)
    ctx.set_tau_rng(_lib.RNG_MT19937)


# ---------------------------------------------------------------- A1 tau sweep
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "tau_sweep_*.npz"))))
def test_tau_sweep_golden(ctx, path):
    z = np.load(path)
    _load(ctx, z["counts"], z["tau_in"], z["gamma"], z["eta"], int(z["mt_seed"]))
    n, logp = ctx.sample_tau(want_logp=True)
    tau, _, _ = ctx.get_state()
    assert np.array_equal(tau, z["tau_out"])                      # bit-exact integer outcome
    # nchange counts (v,g) pairs whose base changed = the net difference (one draw per pair per sweep)
    assert n == int((np.argmax(z["tau_in"], 2) != np.argmax(z["tau_out"], 2)).sum())
    # fp tolerance: lane-parallel summation order + device log vs libm
    np.testing.assert_allclose(logp, z["logp"], rtol=1e-12, atol=1e-8)


@pytest.mark.parametrize("V,S,G", [(1000, 16, 5), (1500, 64, 8), (257, 96, 12), (300, 33, 4), (300, 20, 2),
                                   (64, 130, 3), (50, 300, 6), (7, 1, 1), (1, 64, 8), (200, 64, 16), (100, 8, 32)])
def test_tau_sweep_vs_oracle(ctx, V, S, G):
    counts, _, _ = synth_counts(V, S, max(G, 2), seed=V + S)
    tau, gamma, eta = random_state(V, S, G, seed=G)
    _load(ctx, counts, tau, gamma, eta, mt_seed=4242)
    mt = cbind.MT19937(4242)
    ref = tau.copy()
    for sweep in range(3):                                        # the MT19937 stream carries over
        n_ref, logp_ref = cbind.sample_tau_u(ref, gamma, eta, counts, mt.uniform(V * G), want_logp=True)
        n, logp = ctx.sample_tau(want_logp=True)
        got, _, _ = ctx.get_state()
        assert n == n_ref
        assert np.array_equal(got, ref)
        np.testing.assert_allclose(logp, logp_ref, rtol=1e-12, atol=1e-8)


@pytest.mark.parametrize("kind", ["low", "rare", "floor"])
@pytest.mark.parametrize("V,S,G,spare", [(3000, 64, 8, (2, 5, 7)), (2000, 96, 12, (0, 3, 4, 8, 10, 11)), (1500, 16, 5, (1, 4)),
                                         (600, 200, 6, (0, 5)), (400, 300, 4, (2,)), (500, 40, 3, (0,)), (800, 130, 7, (3, 6)),
                                         (1000, 64, 8, (0, 1, 2, 3, 4, 5, 6))])
def test_tau_sweep_of_an_overfitted_chain_vs_oracle(ctx, V, S, G, spare, kind):
    """A chain with more haplotypes than the table has strains keeps the spare ones at low abundance -- gamma ~ 1e-2 early on
    ("low": races of 8 ... 64 nats, scripts/dbg/flat_diag.py), ~ 1e-3 once the chain has settled ("rare": near-ties,
    scripts/dbg/chain_fp64.py) -- or at the floor (gamma = epsilon = 1e-6 in every sample, HaploSNP_Sampler.py:271-273).  The first
    kind is settled by the screening pass from the totals AND the uniform (dsm_device.h: screen_certify: few steps left to fp64); the
    near-ties of the other two go to the fp64 code.  Either way the draws are the oracle's, screens on or off."""
    live = [g for g in range(G) if g not in spare]
    counts, _, _ = synth_counts(V, S, max(len(live), 2), seed=V + S + G)
    tau, gamma, eta = random_state(V, S, G, seed=G + 100)
    gamma = gamma.copy()
    rng = np.random.default_rng(V + G)
    gamma[:, list(spare)] = 1.0e-6 if kind == "floor" else rng.uniform(2.0e-4, 2.0e-3, size=(S, len(spare))) if kind == "rare" else \
        rng.uniform(1.0e-2, 4.0e-2, size=(S, len(spare)))
    gamma[:, live] *= ((1.0 - gamma[:, list(spare)].sum(axis=1)) / gamma[:, live].sum(axis=1))[:, None]
    gamma = np.ascontiguousarray(gamma / gamma.sum(axis=1)[:, None])
    n_sw = 4
    gs, es = np.ascontiguousarray(np.broadcast_to(gamma, (n_sw,) + gamma.shape)), np.ascontiguousarray(np.broadcast_to(eta, (n_sw, 4, 4)))
    _load(ctx, counts, tau, gamma, eta, mt_seed=777)
    ctx.sweep_stats(reset=True)
    mt = cbind.MT19937(777)
    ref = tau.copy()
    for sweep in range(2):                                        # single sweeps (dsm_ctx_sample_tau), screened
        n_ref = cbind.sample_tau_u(ref, gamma, eta, counts, mt.uniform(V * G))
        n = ctx.sample_tau()
        got, _, _ = ctx.get_state()
        assert n == n_ref and np.array_equal(got, ref)
    ctx.update_tau(gs, es)                                        # ... and the tau-only loop, whose finalize step keeps the counts
    for it in range(n_sw):
        cbind.sample_tau_u(ref, gamma, eta, counts, mt.uniform(V * G))
        assert np.array_equal(ctx.get_tau_at(it), ref)
    steps, exact = ctx.sweep_stats()
    assert steps > 0
    if kind != "floor" and len(live) > 1 and S >= 40 and (kind == "rare" or len(spare) * 0.04 < 0.5):   # (from a random tau the races are wide)
        assert exact / steps < 0.25, (kind, steps, exact)         # (round 3's rule -- gaps above 64 nats only -- left len(spare) / G of them)
    # the same sweeps with the screen switched off: the same haplotypes
    _load(ctx, counts, tau, gamma, eta, mt_seed=777)
    ctx.set_tau_screen(False)
    try:
        for sweep in range(2):
            ctx.sample_tau()
        ctx.update_tau(gs, es)
        assert np.array_equal(ctx.get_tau_at(n_sw - 1), ref)
    finally:
        ctx.set_tau_screen(True)


@pytest.mark.parametrize("V,S,G,spare,scale", [(3000, 64, 8, (2, 5, 7), 1e-3), (2000, 96, 12, (0, 3, 4, 8, 10, 11), 1e-3), (1500, 16, 5, (1, 4), 2e-3),
                                               (600, 200, 6, (0, 5), 5e-4), (400, 300, 4, (2,), 1e-3), (800, 130, 7, (3, 6), 1e-2),
                                               (1200, 40, 6, (0, 1, 2, 3), 1e-6), (900, 500, 3, (1,), 1e-3)])
def test_gibbs_loop_with_the_neartie_sweep_is_the_same_chain(ctx, V, S, G, spare, scale):
    """The Gibbs loop's sweep has a second instantiation with a screen on the DIFFERENCES of the candidates for the steps of haplotypes
    that are rare in every sample (tau_kernel_nt; dsm_ctx_set_tau_neartie).  Forced on, forced off or chosen by the chain's abundances,
    the chain is the same chain bit for bit -- haplotypes of every iteration, gamma / eta traces, ll / lp, MAP record -- and equals the
    oracle's sweeps; and the screen does settle the near-ties (fewer steps left to fp64 than without it)."""
    live = [g for g in range(G) if g not in spare]
    counts, tau_true, gamma_true = synth_counts(V, S, max(len(live), 2), seed=V + G)
    rng = np.random.default_rng(V + S)
    # a settled over-fitted state: the live haplotypes carry the generating tau, the spare ones random bases at low abundance
    tau_idx = rng.integers(0, 4, size=(V, G)).astype(np.uint8)
    tau_idx[:, live] = tau_true[:, :len(live)]
    tau = cbind.idx_to_onehot(tau_idx)
    gamma = np.full((S, G), 0.0)
    gamma[:, list(spare)] = rng.uniform(0.2 * scale, 2.0 * scale, size=(S, len(spare)))
    gamma[:, live] = gamma_true[:, :len(live)] * (1.0 - gamma[:, list(spare)].sum(axis=1))[:, None] / gamma_true[:, :len(live)].sum(axis=1)[:, None]
    gamma = np.ascontiguousarray(gamma / gamma.sum(axis=1)[:, None])
    eta = 0.96 * np.eye(4) + 0.01
    n_it = 5
    runs = {}
    for mode in (0, 1, -1):
        _load(ctx, counts, tau, gamma, eta, mt_seed=4711)
        ctx.seed(4711, ctr_seed=0xFEEDFACE)
        ctx.force_stats_spec(2)
        ctx.set_tau_neartie(mode)
        ctx.sweep_stats(reset=True)
        try:
            ctx.gibbs_update(n_it)
        finally:
            ctx.set_tau_neartie(-1)
            ctx.force_stats_spec(0)
        tr = ctx.get_trace()
        runs[mode] = dict(tr=tr, taus=[ctx.get_tau_at(i) for i in range(n_it)], star=ctx.get_star(), stats=ctx.sweep_stats())
    for mode in (1, -1):
        for k in ("gamma", "eta", "ll", "lp", "nchange"):
            assert np.array_equal(runs[0]["tr"][k], runs[mode]["tr"][k]), (mode, k)
        assert all(np.array_equal(a, b) for a, b in zip(runs[0]["taus"], runs[mode]["taus"]))
        assert runs[0]["star"]["lp"] == runs[mode]["star"]["lp"] and np.array_equal(runs[0]["star"]["tau"], runs[mode]["star"]["tau"])
    # every sweep of the loop against the oracle, on the GSL stream: tau_it from (tau_{it-1}, gamma_it, eta_{it-1})
    mt = cbind.MT19937(4711)
    ref, eta_prev = tau.copy(), eta
    for it in range(n_it):
        cbind.sample_tau_u(ref, np.ascontiguousarray(runs[1]["tr"]["gamma"][it]), np.ascontiguousarray(eta_prev), counts, mt.uniform(V * G))
        assert np.array_equal(runs[1]["taus"][it], ref), it
        eta_prev = runs[1]["tr"]["eta"][it]
    (st0, ex0), (st1, ex1) = runs[0]["stats"], runs[1]["stats"]
    assert st0 == st1 > 0
    assert ex1 <= ex0 + 0.01 * st1, (ex0, ex1, st1)               # never worse than the totals alone ...
    if S >= 40 and len(live) > 1 and scale <= 2e-3:
        assert ex1 < 0.5 * ex0 + 0.02 * st1, (ex0, ex1, st1)      # ... and the near-ties no longer go to fp64
    if scale <= 2e-3:
        assert runs[-1]["stats"] == runs[1]["stats"]              # the abundances of such a state switch it on by themselves


def test_tau_sweep_zero_counts_and_deep_counts(ctx):
    V, S, G = 40, 16, 3
    counts, _, _ = synth_counts(V, S, G, seed=3)
    counts[::3] = 0                                               # empty variants: uniform conditionals
    counts[1, :, :] = counts[1, :, :] * 40000 + 16777217          # > 2^24: the float cast rounds
    tau, gamma, eta = random_state(V, S, G, seed=1)
    _load(ctx, counts, tau, gamma, eta, mt_seed=9)
    ref = tau.copy()
    n_ref, logp_ref = cbind.sample_tau_u(ref, gamma, eta, counts, cbind.MT19937(9).uniform(V * G), want_logp=True)
    n, logp = ctx.sample_tau(want_logp=True)
    got, _, _ = ctx.get_state()
    assert n == n_ref and np.array_equal(got, ref)
    np.testing.assert_allclose(logp, logp_ref, rtol=1e-12, atol=1e-8)


def test_sampletau_dropin_module(ctx):
    """the legacy four-call interface (sampletau.pyx:21-57) end to end"""
    from desman_amd import sampletau
    z = np.load(os.path.join(GOLDEN, "tau_sweep_V64_S16_G5.npz"))
    t_hip, t_ref = z["tau_in"].copy(), z["tau_in"].copy()
    sampletau.initRNG(); sampletau.setRNG(77)
    cbind.initRNG(); cbind.setRNG(77)
    for _ in range(2):
        n = sampletau.sample_tau(t_hip, z["gamma"], z["eta"], z["counts"])
        n_ref = cbind.sample_tau(t_ref, z["gamma"], z["eta"], z["counts"])
        assert n == n_ref and np.array_equal(t_hip, t_ref)
    sampletau.freeRNG(); cbind.freeRNG()
    with pytest.raises(ValueError):
        sampletau.sample_tau(t_hip.astype(np.int32), z["gamma"], z["eta"], z["counts"])
    with pytest.raises(TypeError):
        sampletau.sample_tau(None, z["gamma"], z["eta"], z["counts"])
    with pytest.raises(ValueError):
        sampletau.sample_tau(np.asfortranarray(t_hip), z["gamma"], z["eta"], z["counts"])
    with pytest.raises(_lib.DesmanHipError):                       # RNG freed: loud, no exit(1)
        sampletau.sample_tau(t_hip, z["gamma"], z["eta"], z["counts"])


def test_legacy_shim_keeps_the_tensor_resident_only_while_it_is_unchanged():
    """the shim skips the 20 MB upload when pointer, shape and a hash of every word match -- and must notice an
    in-place rewrite of the same array (borrowed pointers: c_sample_tau.c:95)"""
    from desman_amd import sampletau
    V, S, G = 500, 16, 4
    counts, _, _ = synth_counts(V, S, G, seed=11)
    tau, gamma, eta = random_state(V, S, G, seed=12)
    t_hip, t_ref = tau.copy(), tau.copy()
    sampletau.initRNG(); sampletau.setRNG(3)
    mt = cbind.MT19937(3)
    for step in range(4):
        if step == 2:
            counts[::2] = counts[::2][:, ::-1].copy()             # same buffer, new contents
        if step == 3:
            counts[7, 3, 1] += 1                                  # a single word
        n = sampletau.sample_tau(t_hip, gamma, eta, counts)
        n_ref = cbind.sample_tau_u(t_ref, gamma, eta, counts, mt.uniform(V * G))
        assert n == n_ref and np.array_equal(t_hip, t_ref), step
    sampletau.freeRNG()


def test_reference_named_c_aliases():
    """c_initRNG / c_setRNG / c_freeRNG / c_sample_tau (sampletau.pyx:10-16 binds these names): same stream, same sweep"""
    lib = _lib.load()
    z = np.load(os.path.join(GOLDEN, "tau_sweep_V64_S16_G5.npz"))
    t_hip, t_ref = z["tau_in"].copy(), z["tau_in"].copy()
    V, G, S = t_hip.shape[0], t_hip.shape[1], z["gamma"].shape[0]
    lib.c_initRNG(); lib.c_setRNG(4711)
    cbind.initRNG(); cbind.setRNG(4711)
    for _ in range(2):
        n = lib.c_sample_tau(t_hip, np.ascontiguousarray(z["gamma"]), np.ascontiguousarray(z["eta"]), z["counts"], V, G, S)
        n_ref = cbind.sample_tau(t_ref, z["gamma"], z["eta"], z["counts"])
        assert n == n_ref and np.array_equal(t_hip, t_ref)
    lib.c_freeRNG(); cbind.freeRNG()
    assert lib.c_sample_tau(t_hip, np.ascontiguousarray(z["gamma"]), np.ascontiguousarray(z["eta"]), z["counts"], V, G, S) == -1


def test_sampletau_thread_local_streams():
    """two worker threads with their own logical GSL streams (desman_amd.chains): each sample_tau draws from ITS
    stream on its own context -- results equal the single-threaded reference for each seed, whatever the interleaving"""
    import threading
    from desman_amd import sampletau
    z = np.load(os.path.join(GOLDEN, "tau_sweep_V64_S16_G5.npz"))
    want = {}
    for seed in (5, 6):
        t = z["tau_in"].copy()
        mt = cbind.MT19937(seed)
        V, G = t.shape[0], t.shape[1]
        ns = [cbind.sample_tau_u(t, z["gamma"], z["eta"], z["counts"], mt.uniform(V * G)) for _ in range(3)]
        want[seed] = (t, ns)
    got, errs = {}, []
    barrier = threading.Barrier(2)

    def work(seed):
        try:
            sampletau.use_thread_local_rng(True)
            sampletau.initRNG(); sampletau.setRNG(seed)
            t = z["tau_in"].copy()
            ns = []
            for _ in range(3):
                barrier.wait(timeout=60)                          # force the two threads to interleave their sweeps
                ns.append(sampletau.sample_tau(t, z["gamma"], z["eta"], z["counts"]))
            got[seed] = (t, ns)
            sampletau.freeRNG()
            sampletau.use_thread_local_rng(False)
        except Exception as e:                                    # noqa: BLE001
            errs.append(e)
            barrier.abort()
    th = [threading.Thread(target=work, args=(s,)) for s in (5, 6)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    for seed in (5, 6):
        assert np.array_equal(got[seed][0], want[seed][0]) and got[seed][1] == want[seed][1]


# ---------------------------------------------------------------- A5 ll / lp
def test_loglik_golden(ctx):
    z = np.load(os.path.join(GOLDEN, "loglik.npz"))
    for i in range(int(z["n"])):
        _load(ctx, z["counts_%d" % i], z["tau_%d" % i], z["gamma_%d" % i], z["eta_%d" % i])
        ll, lp = ctx.loglik()
        assert ll == pytest.approx(float(z["ll_%d" % i]), rel=1e-12)
        assert lp == pytest.approx(float(z["lp_%d" % i]), rel=1e-12)


@pytest.mark.parametrize("V,S,G", [(2000, 64, 8), (500, 96, 12), (333, 20, 3)])
def test_loglik_vs_oracle(ctx, V, S, G):
    counts, _, _ = synth_counts(V, S, G, seed=7)
    tau, gamma, eta = random_state(V, S, G, seed=8)
    _load(ctx, counts, tau, gamma, eta)
    ll, lp = ctx.loglik()
    idx = cbind.onehot_to_idx(tau)
    assert ll == pytest.approx(cbind.loglik(idx, gamma, eta, counts), rel=1e-12)
    assert lp == pytest.approx(cbind.logpost(idx, gamma, eta, counts), rel=1e-12)


# ---------------------------------------------------------------- A2 mu/E sums
@pytest.mark.parametrize("V,S,G", [(300, 16, 5), (1000, 64, 8), (70, 5, 3), (129, 7, 12), (50, 3, 20), (90, 4, 1),
                                   (64, 2, 2), (60, 4, 4), (60, 4, 6), (40, 3, 7), (40, 3, 10), (30, 2, 14),
                                   (30, 2, 16), (20, 2, 24), (20, 2, 28), (20, 2, 32),     # every compiled haplotype count
                                   (40, 3, 9), (40, 3, 11), (30, 2, 13), (30, 2, 18), (20, 2, 22), (20, 2, 30)])   # padded ones
def test_stats_bit_exact_vs_spec(ctx, V, S, G):
    """the mu/E sums against the restated counter-based specification, bit for bit: spec v2 (aggregated
    sampler, oracle/stats_agg.c) where it applies (G <= 16), and the per-read spec v1 (orc_stats_counter)
    both where it is the product path (G > 16) and, forced, on the same shapes."""
    counts, _, _ = synth_counts(V, S, max(G, 2), seed=31)
    counts[5] = 0
    tau, gamma, eta = random_state(V, S, G, seed=32)
    _load(ctx, counts, tau, gamma, eta)
    ctx.seed(1, ctr_seed=0xABCDEF0123456789)
    idx = cbind.onehot_to_idx(tau)
    assert ctx.stats_spec() == 1                                 # small problem: per-read pass by the shape rule
    for force in ((3, 2, 1, 4) if G <= 16 else (0,)):             # the versions of the aggregated specification, and the per-read one
        ctx.force_stats_spec(force)
        assert ctx.stats_spec() == ((2 if G > 8 else 4) if force == 4 else force if force >= 2 else 1)
        for it in (0, 1, 77):
            mu, E = ctx.sample_stats(it)
            mu_ref, E_ref = (cbind.stats_agg(idx, gamma, eta, counts, 0xABCDEF0123456789, it, spec=force) if force >= 2 else
                             cbind.stats_counter(idx, gamma, eta, counts, 0xABCDEF0123456789, it))
            assert np.array_equal(mu, mu_ref) and np.array_equal(E, E_ref)
            assert int(mu.sum()) == int(counts.sum())
            # E[b, :] partitions the reads of observed base b
            assert np.array_equal(E.sum(axis=1), counts.sum(axis=(0, 1)).astype(np.uint64))
    ctx.force_stats_spec(0)


@pytest.mark.parametrize("V,S,G,scale", [(200, 64, 8, 1.0), (120, 96, 12, 1.0), (150, 16, 5, 1.0), (60, 130, 3, 1.0),
                                         (80, 64, 8, 20.0), (40, 7, 16, 1.0),
                                         # deferred lists of every kind longer than one workgroup of the compacted kernel
                                         (400, 64, 8, 10.0),
                                         # few subsets, many positions: the subset table in 4 / 8 copies (stats_ntab_rep)
                                         (800, 20, 2, 1.0), (2500, 12, 3, 1.0), (64, 64, 1, 1.0), (50, 300, 6, 0.05),
                                         # lane groups of 16 / 32 lanes with ragged last chunks and a ragged last wavefront pass
                                         (91, 48, 6, 1.0), (77, 33, 4, 1.0), (53, 100, 5, 1.0), (35, 32, 7, 1.0), (9, 20, 2, 1.0)])
def test_stats_stage1_matches_spec(ctx, V, S, G, scale):
    """stage 1 of spec v2 alone: the subset counts N[s][H] and Esum, on random (burn-in like) and on
    generating (converged) states, shallow and 20x deep data."""
    counts, tau_true, gamma_true = synth_counts(V, S, max(G, 2), seed=33, depth_scale=scale)
    counts[3] = 0
    counts[7, :, :] = counts[7, :, :] * 3000 + 5                  # one very deep variant: chunked inversion
    seed = 0x0123456789ABCDEF
    for state in ("random", "truth"):
        if state == "random" or G < 2:
            tau, gamma, eta = random_state(V, S, G, seed=34)
        else:
            tau, gamma, eta = cbind.idx_to_onehot(tau_true[:, :G]), np.ascontiguousarray(gamma_true[:, :G]), 0.96 * np.eye(4) + 0.01
            gamma = np.ascontiguousarray(gamma / gamma.sum(axis=1, keepdims=True))
        _load(ctx, counts, tau, gamma, eta)
        ctx.seed(1, ctr_seed=seed)
        idx = cbind.onehot_to_idx(tau)
        for spec in (3, 2, 4):
            ctx.force_stats_spec(spec)
            for it in (0, 5):
                nt, E = ctx.debug_stage1(it)
                mu_ref, E_ref, nt_ref = cbind.stats_agg(idx, gamma, eta, counts, seed, it, want_ntab=True, spec=spec)
                assert np.array_equal(E, E_ref)
                assert np.array_equal(nt, nt_ref)
                mu, E2 = ctx.sample_stats(it)
                assert np.array_equal(mu, mu_ref) and np.array_equal(E2, E_ref)
    ctx.force_stats_spec(0)


@pytest.mark.parametrize("V,S,G,scale,biallelic", [(5000, 64, 3, 1.0, False), (3000, 96, 2, 1.0, False), (2000, 16, 4, 1.0, False),
                                                   (1500, 40, 5, 1.0, False), (800, 130, 2, 1.0, False), (1200, 64, 8, 1.0, True),
                                                   (2500, 64, 3, 25.0, False), (1000, 33, 6, 8.0, True), (333, 200, 1, 1.0, False),
                                                   # pooled in LDS first (V >= 4096, G <= 4): 64- and 32-lane position groups, ragged S and V
                                                   (6000, 96, 4, 1.0, False), (4099, 40, 4, 1.0, False), (9001, 100, 2, 1.0, False), (4500, 7, 1, 1.0, False)])
def test_stats_over_tau_words_matches_twin(ctx, V, S, G, scale, biallelic):
    """spec 4: positions that share their packed tau word share one stage-1 cell per sample (pat_rep_kernel / pat_agg_kernel /
    stats_pat_kernel) -- against the twin (cbind.stats_agg(spec=4): counts pooled in the lowest position of the word, then the spec-2
    code), bit for bit: subset counts, Esum, sum_mu; deep tables push pooled items onto the deferred lists (stats_big_kernel reads
    and clears the pooled counts); a pass leaves the pooled counts zero, so passes repeat."""
    counts, tau_true, gamma_true = synth_counts(V, S, max(G, 2), seed=V + G, depth_scale=scale)
    rng = np.random.default_rng(V)
    if biallelic:                                                 # every position has two alleles: at most 12 (2^G - 2) + 4 words
        a0, a1 = rng.integers(0, 4, size=V), rng.integers(1, 4, size=V)
        idx = np.where(rng.random((V, G)) < 0.5, a0[:, None], ((a0 + a1) % 4)[:, None]).astype(np.uint8)
        tau = cbind.idx_to_onehot(idx)
        _, gamma, eta = random_state(V, S, G, seed=5)
    else:
        tau, gamma, eta = random_state(V, S, G, seed=G + 7)
    _load(ctx, counts, tau, gamma, eta)
    seed = 0x00C0FFEE12345678
    ctx.seed(1, ctr_seed=seed)
    idx = cbind.onehot_to_idx(tau)
    n_words = len(np.unique(idx, axis=0))
    assert n_words < V
    ctx.force_stats_spec(4)
    try:
        assert ctx.stats_spec() == 4
        first = None
        for it in (0, 3, 0):
            nt, E = ctx.debug_stage1(it)
            mu_ref, E_ref, nt_ref = cbind.stats_agg(idx, gamma, eta, counts, seed, it, want_ntab=True, spec=4)
            assert np.array_equal(E, E_ref) and np.array_equal(nt, nt_ref)
            mu, E2 = ctx.sample_stats(it)
            assert np.array_equal(mu, mu_ref) and np.array_equal(E2, E_ref)
            assert int(mu.sum()) == int(counts.sum()) and np.array_equal(E.sum(axis=1), counts.sum(axis=(0, 1)).astype(np.uint64))
            if it == 0:
                if first is None:
                    first = mu.copy()
                else:
                    assert np.array_equal(first, mu)
        # another state on the same context: the word table carries nothing over from the pass before
        tau2, _, _ = random_state(V, S, G, seed=G + 99)
        ctx.set_state(tau2, gamma, eta)
        mu, E = ctx.sample_stats(1)
        mu_ref, E_ref = cbind.stats_agg(cbind.onehot_to_idx(tau2), gamma, eta, counts, seed, 1, spec=4)
        assert np.array_equal(mu, mu_ref) and np.array_equal(E, E_ref)
    finally:
        ctx.force_stats_spec(0)


def test_stats_degenerate_eta_and_gamma(ctx):
    """eta with exact zeros (identity) and a gamma column of zeros: weights vanish for whole bases / subsets"""
    V, S, G = 50, 16, 4
    counts, _, _ = synth_counts(V, S, G, seed=35)
    tau, gamma, eta = random_state(V, S, G, seed=36)
    eta = np.eye(4)
    gamma[:, 2] = 0.0
    gamma = np.ascontiguousarray(gamma / gamma.sum(axis=1, keepdims=True))
    _load(ctx, counts, tau, gamma, eta)
    ctx.seed(1, ctr_seed=9)
    idx = cbind.onehot_to_idx(tau)
    for spec in (3, 2, 4):
        ctx.force_stats_spec(spec)
        mu, E = ctx.sample_stats(2)
        ctx.force_stats_spec(0)
        mu_ref, E_ref = cbind.stats_agg(idx, gamma, eta, counts, 9, 2, spec=spec)
        assert np.array_equal(mu, mu_ref) and np.array_equal(E, E_ref)
        assert (mu[:, 2] == 0).all() and int(mu.sum()) == int(counts.sum())


@pytest.mark.parametrize("kind,n,w", [(0, 300, (0.02, 0.98)), (0, 300, (0.98, 0.02)), (0, 300, (0.5, 0.5)), (0, 13, (0.5, 0.1)),
                                      (0, 5000, (0.4, 0.6)), (0, 2 ** 31 - 1, (1e-9, 1.0)), (0, 77, (0.0, 1.0)), (0, 77, (1.0, 0.0)),
                                      (1, 1000, (0.3, 0.7)), (1, 100000, (0.013, 1.0)), (1, 33, (0.5, 0.5)), (1, 2 ** 32 - 1, (0.5, 0.5)),
                                      (1, 4000000000, (1e-8, 1.0)), (1, 3000, (0.99, 0.01)), (1, 17, (0.3, 0.7)),
                                      (2, 300, (0.003, 0.002, 0.99, 0.005)), (2, 9, (0.1, 0.2, 0.3, 0.4)), (2, 500, (0.25, 0.25, 0.3, 0.2)),
                                      (2, 40, (0.0, 0.5, 0.5, 0.0)), (2, 100000, (0.96, 0.01, 0.02, 0.01)), (2, 13, (1e-300, 1.0, 0.0, 1e-300))])
def test_binomial_and_multinomial_samplers_match_spec(ctx, kind, n, w):
    """dsm_binom.h against its restatement in oracle/stats_agg.c, variate by variate (same streams)"""
    nsamp = 20000
    for spec in (3, 2):
        got = ctx.debug_binom(kind, n, w, 0xFEEDFACE12345678, nsamp, spec=spec)
        if kind == 2:
            ref = cbind.mult4_test(n, w, 0xFEEDFACE12345678, nsamp, spec=spec)
            assert (got.sum(axis=1) == n).all()
        else:
            ref = cbind.binom_test(kind, n, w[0], w[1], 0xFEEDFACE12345678, nsamp, spec=spec)
        assert np.array_equal(got, ref), spec


@pytest.mark.parametrize("name", sorted(LAW_CASES))
def test_stats_law_matches_reference_sampleMu(spec_ctx, name):
    """the counter-based device draws of (sum_mu, Esum) and the reference's two-stage numpy draw
    (HaploSNP_Sampler.py:284-309, oracle/ref_numpy.py: sample_mu) have the same law: two-sample chi-square on every
    per-(s,g) marginal of sum_mu and every entry of Esum (2000 draws a side), means and variances; cases: the small
    shape, G = 10 and 12 (stage 2 as its own launch), x 15 depth (BTRS, deferred lists, stats_big_kernel), and a
    converged state with eta ~ 0.97 I (rare-outcome inversion).  tests/test_law_cpu.py does the same for the oracle twins."""
    ctx = spec_ctx
    counts, tau, gamma, eta = law_case(name)
    _load(ctx, counts, tau, gamma, eta)
    ctx.seed(1, ctr_seed=5)
    idx = cbind.onehot_to_idx(tau)
    n = 2000
    mus, es = [], []
    for it in range(n):
        mu, E = ctx.sample_stats(it)
        mus.append(mu.astype(np.int64)); es.append(E.astype(np.int64))
    mus, es = np.array(mus), np.array(es)
    assert (mus.sum(axis=2) == counts.sum(axis=(0, 2))[None, :]).all()
    mu_r, E_r = reference_draws(name, n)
    assert_same_law(mus, es, mu_r, E_r, name)
    e_mu, v_mu, e_E = cbind.stats_expect(idx, gamma, eta, counts)
    z = (mus.mean(axis=0) - e_mu) / np.sqrt(v_mu / n + 1e-12)
    assert np.abs(z).max() < 5.0
    np.testing.assert_allclose(es.mean(axis=0), e_E, rtol=0.02, atol=5.0 * np.sqrt(e_E.max() / n) + 0.5)


# ---------------------------------------------------------------- A3/A4 Dirichlet draws
def test_gamma_eta_draws(ctx):
    V, S, G = 20, 64, 8
    counts, _, _ = synth_counts(V, S, G, seed=50)
    tau, gamma, eta = random_state(V, S, G, seed=51)
    _load(ctx, counts, tau, gamma, eta)
    ctx.seed(1, ctr_seed=11)
    rng = np.random.default_rng(0)
    sum_mu = rng.integers(0, 50, size=(S, G)).astype(np.uint64)
    sum_mu[0, :] = 0                                              # shape 0.1 everywhere: clamp path
    sum_mu[1, 1:] = 0; sum_mu[1, 0] = 10 ** 7
    # asymmetric on purpose: eta[a,:] ~ Dir(delta + Esum[:,a]) (HaploSNP_Sampler.py:281) -- a transposed read
    # of E[observed,true] would fail the mean test below
    esum = (np.eye(4) * 50000 + np.array([[0, 300, 4000, 90], [2500, 0, 40, 700], [60, 5000, 0, 1500],
                                          [900, 10, 3000, 0]])).astype(np.uint64)
    n = 600
    g_acc = np.zeros((S, G)); g2 = np.zeros((S, G)); e_acc = np.zeros((4, 4))
    for it in range(n):
        g, e = ctx.draw_gamma_eta(it, sum_mu, esum)
        np.testing.assert_allclose(g.sum(axis=1), 1.0, rtol=1e-12)
        np.testing.assert_allclose(e.sum(axis=1), 1.0, rtol=1e-12)
        assert g.min() >= 1e-6 / (1.0 + G * 1e-6) * (1 - 1e-12) and (e > 0).all()
        g_acc += g; g2 += g * g; e_acc += e
    a = 0.1 + sum_mu.astype(np.float64)
    a0 = a.sum(axis=1, keepdims=True)
    mean = a / a0
    var = mean * (1 - mean) / (a0 + 1)
    rows = np.arange(2, S)                                        # rows 0,1 are dominated by the clamp
    z = (g_acc[rows] / n - mean[rows]) / np.sqrt(var[rows] / n)
    assert np.abs(z).max() < 4.5
    np.testing.assert_allclose(g2[rows] / n - (g_acc[rows] / n) ** 2, var[rows], rtol=0.35, atol=1e-6)
    d = 0.1 + esum.T.astype(np.float64)                           # eta[a,:] ~ Dir(delta + Esum[:,a])
    np.testing.assert_allclose(e_acc / n, d / d.sum(axis=1, keepdims=True), rtol=2e-3, atol=2e-4)
    wrong = 0.1 + esum.astype(np.float64)
    assert np.abs(e_acc / n - wrong / wrong.sum(axis=1, keepdims=True)).max() > 0.02      # the test discriminates
    # determinism: same (seed, iter) -> same draw
    g1, e1 = ctx.draw_gamma_eta(5, sum_mu, esum)
    g1b, e1b = ctx.draw_gamma_eta(5, sum_mu, esum)
    assert np.array_equal(g1, g1b) and np.array_equal(e1, e1b)
    # the deterministic tail of sampleGamma (HaploSNP_Sampler.py:271-273)
    assert np.allclose(g1, rn.clamp_renorm_gamma(g1), rtol=1e-12)


@pytest.mark.parametrize("S,G", [(1, 1), (1, 8), (16, 8), (64, 8), (64, 32), (16, 1), (512, 8), (512, 32), (96, 12)])
def test_dirichlet_draws_match_spec(ctx, S, G):
    """A3/A4 value by value against the restated draw specification (oracle: orc_dirichlet_counter): Philox
    counter layout, Box-Muller, Marsaglia-Tsang, shape < 1 boost, clamp / renormalise.  Tolerance 1e-13
    relative: the device's log / cos / pow and glibc's differ in the last bits, nothing else may."""
    V = 4
    counts, _, _ = synth_counts(V, S, max(G, 2), seed=52)
    tau, gamma, eta = random_state(V, S, G, seed=53)
    _load(ctx, counts, tau, gamma, eta)
    seed = 0x1234ABCD5678EF01
    ctx.seed(1, ctr_seed=seed)
    rng = np.random.default_rng(S * 100 + G)
    sum_mu = rng.integers(0, 3000, size=(S, G)).astype(np.uint64)
    sum_mu[0, :] = 0                                              # every shape = alpha = 0.1: boost + clamp path
    if S > 1:
        sum_mu[1, :] = 0; sum_mu[1, 0] = 10 ** 7                  # one huge shape next to 0.1s
    if S > 2:
        sum_mu[2, :] = 10 ** 7
    esum = rng.integers(0, 9000, size=(4, 4)).astype(np.uint64)   # asymmetric
    esum[3, :] = 0; esum[3, 3] = 10 ** 7
    for it in (0, 3, 1000):
        g, e = ctx.draw_gamma_eta(it, sum_mu, esum)
        g_ref, e_ref, _ = cbind.dirichlet_counter(sum_mu, esum, seed, it)
        np.testing.assert_allclose(g, g_ref, rtol=1e-13, atol=0)
        np.testing.assert_allclose(e, e_ref, rtol=1e-13, atol=0)


def test_gibbs_chain_recovers_asymmetric_eta(spec_ctx):
    """chain-level check of the E[observed,true] -> eta[true,:] indexing (HaploSNP_Sampler.py:275-281):
    data generated with a strongly asymmetric error matrix; the posterior mean of eta must be that matrix,
    not its transpose."""
    ctx = spec_ctx
    V, S, G = 400, 16, 3
    eta_true = np.array([[0.90, 0.07, 0.02, 0.01],
                         [0.01, 0.96, 0.01, 0.02],
                         [0.05, 0.01, 0.93, 0.01],
                         [0.01, 0.01, 0.10, 0.88]])
    counts, tau_true, gamma_true = synth_counts(V, S, G, seed=71, eta=eta_true)
    tau0 = cbind.idx_to_onehot(tau_true)
    _load(ctx, counts, tau0, np.ascontiguousarray(gamma_true), 0.96 * np.eye(4) + 0.01, mt_seed=6)
    ctx.gibbs_update(60)
    ctx.gibbs_update(200)
    eta_mean = ctx.get_trace()["eta"].mean(axis=0)
    np.testing.assert_allclose(eta_mean, eta_true, atol=0.012)
    assert np.abs(eta_mean - eta_true.T).max() > 0.04


# ---------------------------------------------------------------- A6 full iteration
@pytest.mark.parametrize("spec", [3, 2, 1, 4])
@pytest.mark.parametrize("V,S,G,n_iter", [(400, 16, 5, 12), (600, 64, 8, 8), (150, 96, 3, 6), (200, 20, 11, 4), (900, 10, 2, 4)])
def test_gibbs_update_is_self_consistent_with_oracle(ctx, V, S, G, n_iter, spec):
    """every piece of every iteration of the device loop against the oracle, in the reference's order
    (HaploSNP_Sampler.py:341-358): the mu/E sums + gamma/eta draws from the restated counter-based specs (spec 2:
    stage 2 runs fused inside the Dirichlet launch; spec 1: per-read pass), the tau sweep bit for bit on the GSL
    stream, ll / lp, MAP record, tau sum."""
    counts, _, _ = synth_counts(V, S, G, seed=60)
    tau0, gamma0, eta0 = random_state(V, S, G, seed=61)
    _load(ctx, counts, tau0, gamma0, eta0, mt_seed=123)
    cseed = 0x5EEDC0DE0000 + spec
    ctx.seed(123, ctr_seed=cseed)
    ctx.force_stats_spec(spec)
    ll0, lp0 = ctx.loglik()
    ctx.gibbs_update(n_iter)
    ctx.force_stats_spec(0)
    tr = ctx.get_trace()
    g_prev, e_prev, t_prev = gamma0, eta0, tau0
    for it in range(n_iter):                                      # A2 + A3 + A4 inside the loop; counter = iteration index
        args = (cbind.onehot_to_idx(t_prev), np.ascontiguousarray(g_prev), np.ascontiguousarray(e_prev), counts, cseed, it)
        mu, E = cbind.stats_agg(*args, spec=spec) if spec >= 2 else cbind.stats_counter(*args)
        g_ref, e_ref, _ = cbind.dirichlet_counter(mu, E, cseed, it)
        np.testing.assert_allclose(tr["gamma"][it], g_ref, rtol=1e-13, atol=0)
        np.testing.assert_allclose(tr["eta"][it], e_ref, rtol=1e-13, atol=0)
        g_prev, e_prev, t_prev = tr["gamma"][it], tr["eta"][it], ctx.get_tau_at(it)
    mt = cbind.MT19937(123)
    tau_prev, eta_prev = tau0.copy(), eta0.copy()
    lps = [lp0]
    tau_sum = np.zeros_like(tau0)
    for it in range(n_iter):
        tau_it = ctx.get_tau_at(it)
        # the sweep used gamma_new and eta_old (HaploSNP_Sampler.py:342-347) and the GSL-order uniforms
        ref = tau_prev.copy()
        n_ref = cbind.sample_tau_u(ref, np.ascontiguousarray(tr["gamma"][it]), eta_prev, counts, mt.uniform(V * G))
        assert np.array_equal(tau_it, ref) and tr["nchange"][it] == n_ref
        idx = cbind.onehot_to_idx(tau_it)
        g_it, e_it = np.ascontiguousarray(tr["gamma"][it]), np.ascontiguousarray(tr["eta"][it])
        assert tr["ll"][it] == pytest.approx(cbind.loglik(idx, g_it, e_it, counts), rel=1e-12)
        assert tr["lp"][it] == pytest.approx(cbind.logpost(idx, g_it, e_it, counts), rel=1e-12)
        np.testing.assert_allclose(g_it.sum(axis=1), 1.0, rtol=1e-12)
        lps.append(tr["lp"][it]); tau_sum += tau_it
        tau_prev, eta_prev = tau_it, e_it
    tau_f, gamma_f, eta_f = ctx.get_state()
    assert np.array_equal(tau_f, tau_prev) and np.array_equal(gamma_f, tr["gamma"][-1]) \
        and np.array_equal(eta_f, tr["eta"][-1])
    assert np.array_equal(ctx.get_tau_sum(), tau_sum)
    star = ctx.get_star()
    k = int(np.argmax(lps))                                       # first strict maximum, entry state = slot 0
    assert star["lp"] == lps[k]
    if k == 0:
        assert np.array_equal(star["tau"], tau0) and np.array_equal(star["gamma"], gamma0)
    else:
        assert np.array_equal(star["tau"], ctx.get_tau_at(k - 1)) and np.array_equal(star["gamma"], tr["gamma"][k - 1]) \
            and np.array_equal(star["eta"], tr["eta"][k - 1]) and star["it"] == k - 1


def test_gibbs_chain_recovers_truth(spec_ctx):
    """statistical parity at chain level: on well-identified synthetic data the
    sampler finds the generating haplotypes/abundances (up to relabelling)."""
    ctx = spec_ctx
    V, S, G = 300, 24, 3
    counts, tau_true, gamma_true = synth_counts(V, S, G, seed=70)
    rs = np.random.RandomState(5)
    gamma0, tau0 = rn.sampler_ctor_draws(rs, V, S, G)
    eta0 = 0.96 * np.eye(4) + 0.01
    _load(ctx, counts, tau0, gamma0, eta0, mt_seed=5)
    ctx.gibbs_update(60)
    ctx.gibbs_update(60)
    star = ctx.get_star()
    idx = np.argmax(star["tau"], axis=2)
    import itertools
    best = min(itertools.permutations(range(G)), key=lambda p: (idx[:, list(p)] != tau_true).sum())
    assert (idx[:, list(best)] != tau_true).mean() < 0.02
    g_mean = ctx.get_trace()["gamma"].mean(axis=0)
    np.testing.assert_allclose(g_mean[:, list(best)], gamma_true, atol=0.03)
    eta_mean = ctx.get_trace()["eta"].mean(axis=0)
    np.testing.assert_allclose(eta_mean, 0.96 * np.eye(4) + 0.01, atol=0.01)


def test_update_tau_path(ctx):
    """updateTau (HaploSNP_Sampler.py:383-407): tau-only sweeps over stored traces"""
    V, S, G, n = 200, 16, 4, 5
    counts, _, _ = synth_counts(V, S, G, seed=80)
    tau0, gamma0, eta0 = random_state(V, S, G, seed=81)
    rng = np.random.default_rng(1)
    gs = np.ascontiguousarray(rng.dirichlet(np.ones(G), size=(n, S)))
    es = np.ascontiguousarray(np.stack([random_state(1, 1, 1, seed=k)[2] for k in range(n)]))
    _load(ctx, counts, tau0, gamma0, eta0, mt_seed=99)
    ctx.update_tau(gs, es)
    tr = ctx.get_trace()
    mt = cbind.MT19937(99)
    ref = tau0.copy()
    lp_best, tau_best = cbind.logpost(cbind.onehot_to_idx(ref), gs[0], es[0], counts), ref.copy()
    for it in range(n):
        cbind.sample_tau_u(ref, gs[it], es[it], counts, mt.uniform(V * G))
        assert np.array_equal(ctx.get_tau_at(it), ref)
        lp = cbind.logpost(cbind.onehot_to_idx(ref), gs[it], es[it], counts)
        assert tr["lp"][it] == pytest.approx(lp, rel=1e-12)
        assert tr["ll"][it] == pytest.approx(cbind.loglik(cbind.onehot_to_idx(ref), gs[it], es[it], counts), rel=1e-12)
        if lp > lp_best:
            lp_best, tau_best = lp, ref.copy()
    star = ctx.get_star()
    assert np.array_equal(star["tau"], tau_best) and star["lp"] == pytest.approx(lp_best, rel=1e-12)


# ---------------------------------------------------------------- A8-A12 NMFT
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "nmft_*.npz"))))
def test_nmft_golden(ctx, path):
    z = np.load(path)
    counts, G = z["counts"], int(z["G"])
    ctx.set_counts(counts)
    for k in (1, 10, 100):
        ctx.nmft_set(z["tau_raw"], z["gamma_raw"])
        n, tr = ctx.nmft_factorize(max_iter=k, min_change=0.0)
        # min_change = 0 stops only at an exact fixed point (G = 1 reaches one after 2 updates)
        assert n == k or (n < k and tr[-1] == tr[-2])
        tau, gam = ctx.nmft_get()
        # fp tolerance: reduction order differs from BLAS; drift is rounding-level (north star: 1e-5 rel)
        np.testing.assert_allclose(tau, z["tau_%d" % k], rtol=1e-7, atol=1e-13)
        np.testing.assert_allclose(gam, z["gamma_%d" % k], rtol=1e-7, atol=1e-13)
        assert tr[0] == pytest.approx(float(z["div0"]), rel=1e-11)
        assert tr[-1] == pytest.approx(float(z["div_%d" % k]), rel=1e-9)
        assert ctx.nmft_objective() == pytest.approx(float(z["div_%d" % k]), rel=1e-9)
    assert np.array_equal(ctx.nmft_get_tau(), z["get_tau_100"])
    # factorize() with the reference's stopping rule, max_iter = 300
    ctx.nmft_set(z["tau_raw"], z["gamma_raw"])
    tc, gc = z["tau_raw"].copy(), z["gamma_raw"].copy()
    n_ref, tr_ref = cbind.nmft_factorize(z["F"].copy(), tc, gc, max_iter=300)
    n, tr = ctx.nmft_factorize(max_iter=300, min_change=1e-5)
    assert n == n_ref
    tau, gam = ctx.nmft_get()
    np.testing.assert_allclose(tau, z["fact_tau"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(gam, z["fact_gamma"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(tr, tr_ref, rtol=1e-9)
    assert np.array_equal(ctx.nmft_get_tau(), z["fact_get_tau"])
    # factorize_tau: gamma fixed, no _adjustment
    ctx.nmft_set(z["ft_tau_raw"], z["fact_gamma"])
    ctx.nmft_factorize(max_iter=50, min_change=1e-5, fix_gamma=True)
    tau, gam = ctx.nmft_get()
    np.testing.assert_allclose(tau, z["ft_tau"], rtol=1e-7, atol=1e-13)
    assert np.array_equal(gam, z["fact_gamma"])
    assert np.array_equal(ctx.nmft_get_tau(), z["ft_get_tau"])


@pytest.mark.parametrize("V,S,G", [(700, 64, 8), (300, 96, 12), (500, 16, 5), (40, 300, 4)])
def test_nmft_vs_oracle(ctx, V, S, G):
    counts, _, _ = synth_counts(V, S, G, seed=90)
    tau, gam = rn.nmft_random_initialize(np.random.RandomState(1), V, S, G)
    ctx.set_counts(counts)
    ctx.nmft_set(tau, gam)
    F = cbind.nmft_freq(counts)
    tc, gc = tau.copy(), gam.copy()
    n_ref, tr_ref = cbind.nmft_factorize(F, tc, gc, max_iter=40, min_change=1e-5)
    n, tr = ctx.nmft_factorize(max_iter=40, min_change=1e-5)
    assert n == n_ref
    np.testing.assert_allclose(tr, tr_ref, rtol=1e-9)
    t, g = ctx.nmft_get()
    np.testing.assert_allclose(t, tc, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(g, gc, rtol=1e-6, atol=1e-12)
    assert np.array_equal(ctx.nmft_get_tau(), cbind.idx_to_onehot(cbind.nmft_get_tau(tc, G)))


def test_factorize_tau_after_nmft_set_with_a_larger_rank_on_one_context(ctx):
    """One context, nmft_set(G = 3) + factorize_tau, then nmft_set(G = 8) + factorize_tau: the fused pass's second tau buffer
    follows the rank (it kept the first call's size: device writes past its end).  Both runs equal the oracle's."""
    V, S = 600, 64
    counts, _, _ = synth_counts(V, S, 8, seed=91)
    ctx.set_counts(counts)
    F = cbind.nmft_freq(counts)
    for G in (3, 8, 2):
        tau, gam = rn.nmft_random_initialize(np.random.RandomState(G), V, S, G)
        ctx.nmft_set(tau, gam)
        tc, gc = tau.copy(), gam.copy()
        n_ref, tr_ref = cbind.nmft_factorize_tau(F, tc, gc, max_iter=31, min_change=1e-5)
        n, tr = ctx.nmft_factorize(max_iter=31, min_change=1e-5, fix_gamma=True)
        assert n == n_ref
        np.testing.assert_allclose(tr, tr_ref, rtol=1e-9)
        t, g = ctx.nmft_get()
        np.testing.assert_allclose(t, tc, rtol=1e-6, atol=1e-12)
        assert np.array_equal(g, gam)


def test_chain_posterior_matches_reference_sampler_in_law(spec_ctx):
    """T1 parity at chain level: the HIP chain (counter-based mu/E, gamma, eta draws) and the oracle's
    RandomState-exact restatement of the reference's update() target the same posterior: posterior means of
    gamma, eta and the deviance agree within Monte-Carlo error when both start from the generating state."""
    ctx = spec_ctx
    V, S, G = 60, 8, 3
    counts, tau_true, gamma_true = synth_counts(V, S, G, seed=202)
    tau0 = cbind.idx_to_onehot(tau_true)
    eta0 = 0.96 * np.eye(4) + 0.01
    # reference law: python-level loops with numpy's RandomState + the C tau sweep
    rs = np.random.RandomState(11)
    cbind.initRNG(); cbind.setRNG(11)
    burn = rn.gibbs_update(rs, tau0, gamma_true, eta0, counts, 30, cbind.sample_tau)
    ref = rn.gibbs_update(rs, burn["tau"], burn["gamma"], burn["eta"], counts, 160, cbind.sample_tau)
    cbind.freeRNG()
    # HIP chain
    _load(ctx, counts, tau0, np.ascontiguousarray(gamma_true), eta0, mt_seed=12)
    ctx.gibbs_update(200)
    ctx.gibbs_update(3000)
    tr = ctx.get_trace()
    g_ref, g_hip = ref["trace"]["gamma"], tr["gamma"]
    se = np.sqrt(g_ref.var(axis=0) / 40.0 + g_hip.var(axis=0) / 300.0) + 2e-3     # autocorrelation-padded
    assert (np.abs(g_ref.mean(axis=0) - g_hip.mean(axis=0)) < 5.0 * se).all()
    e_ref, e_hip = ref["trace"]["eta"], tr["eta"]
    se_e = np.sqrt(e_ref.var(axis=0) / 40.0 + e_hip.var(axis=0) / 300.0) + 1e-3
    assert (np.abs(e_ref.mean(axis=0) - e_hip.mean(axis=0)) < 5.0 * se_e).all()
    dev_ref, dev_hip = -2 * ref["trace"]["ll"], -2 * tr["ll"]
    se_d = np.sqrt(dev_ref.var() / 40.0 + dev_hip.var() / 300.0)
    assert abs(dev_ref.mean() - dev_hip.mean()) < 5.0 * se_d + 1.0
    # and the haplotypes both chains settle on are the generating ones
    assert (np.argmax(ctx.get_star()["tau"], axis=2) != tau_true).mean() < 0.05
    assert (np.argmax(ref["star"]["tau"], axis=2) != tau_true).mean() < 0.05


def test_per_read_draws_have_multinomial_mean_and_variance(spec_ctx):
    """distributional check of the xoshiro128+/Philox per-read draws beyond the mean: over many
    iterations the per-haplotype totals of one deep cell have the multinomial variance n p (1-p) and
    the right pairwise covariance -n p_g p_h (correlated or biased words would inflate / deflate them)."""
    ctx = spec_ctx
    V, S, G = 4, 2, 4
    counts = np.zeros((V, S, 4), dtype=np.int64)
    counts[0, 0, 0] = 20000                                     # one deep item dominates; the others stay small
    counts[1:, :, :] = 3
    tau, gamma, eta = random_state(V, S, G, seed=9)
    gamma[0] = [0.4, 0.3, 0.2, 0.1]
    _load(ctx, counts, tau, gamma, eta)
    ctx.seed(1, ctr_seed=2024)
    n_it = 1500
    draws = np.zeros((n_it, G))
    small = counts[:, 0, :].sum() - 20000
    for it in range(n_it):
        mu, _ = ctx.sample_stats(it)
        draws[it] = mu[0]
    idx = np.argmax(tau[0], axis=1)
    w = gamma[0] * eta[idx, 0]
    p_ = w / w.sum()
    n = 20000
    mean = draws.mean(axis=0)
    assert np.abs(mean - n * p_).max() < 5 * np.sqrt(n * 0.25 / n_it) + small
    var = draws.var(axis=0, ddof=1)
    want = n * p_ * (1 - p_)
    # sampling error of a variance estimate ~ want * sqrt(2 / n_it); the small items add at most `small`
    assert (np.abs(var - want) < 6 * want * np.sqrt(2.0 / n_it) + 2 * small).all()
    cov01 = np.cov(draws[:, 0], draws[:, 1])[0, 1]
    assert abs(cov01 + n * p_[0] * p_[1]) < 6 * np.sqrt(want[0] * want[1] / n_it) + 2 * small


def test_reference_stream_chain_with_device_sweep_is_the_reference_trajectory():
    """Module-swap use: the reference's own update() loop -- here its RandomState-exact
    restatement oracle.ref_numpy.gibbs_update, pinned to the imported reference on the CPU -- with
    sampletau.sample_tau replaced by the device sweep reproduces the reference's whole trajectory:
    gamma / eta stores, tau, tau_star, MAP log-posterior identical, ll to 1e-13."""
    from desman_amd import sampletau
    from oracle import ref_numpy as rn
    z = np.load(os.path.join(GOLDEN, "gibbs_pieces.npz"))
    G, seed = int(z["G"]), int(z["seed"])
    counts = z["counts"]
    rs = np.random.RandomState(seed)
    gamma0, tau0 = rn.sampler_ctor_draws(rs, counts.shape[0], counts.shape[1], G)
    sampletau.initRNG(); sampletau.setRNG(seed)
    r = rn.gibbs_update(rs, tau0, gamma0, z["eta0"], counts, 4, sampletau.sample_tau)
    sampletau.freeRNG()
    np.testing.assert_allclose(r["trace"]["ll"], z["ll_store"], rtol=1e-13)
    assert np.array_equal(r["trace"]["gamma"], z["gamma_store"])
    assert np.array_equal(r["trace"]["eta"], z["eta_store"])
    assert np.array_equal(r["tau"], z["tau_final"])
    assert np.array_equal(r["star"]["tau"], z["tau_star"])
    assert r["star"]["lp"] == pytest.approx(float(z["lp_star"]), rel=1e-13)


def test_random_shapes_full_iteration_and_stage1():
    """seeded sweep over odd shapes (V, S, G, iterations drawn at random: ragged lane groups, single samples / haplotypes,
    chunk boundaries of the sweep-word generator): the whole-iteration oracle composition under both mu/E
    specifications, and stage 1 of the aggregated pass on shallow / deep data"""
    rng = np.random.default_rng(20260929)
    for _ in range(10):
        V, S, G, n_iter = int(rng.integers(1, 300)), int(rng.integers(1, 140)), int(rng.integers(1, 13)), int(rng.integers(1, 12))
        for spec in (2, 1):
            c = _lib.Context(0)
            try:
                test_gibbs_update_is_self_consistent_with_oracle(c, V, S, G, n_iter, spec)
            finally:
                c.close()
    for _ in range(8):
        V, S, G = int(rng.integers(8, 200)), int(rng.integers(1, 150)), int(rng.integers(1, 17))
        c = _lib.Context(0)
        try:
            test_stats_stage1_matches_spec(c, V, S, G, float(rng.choice([0.05, 1.0, 20.0])))
        finally:
            c.close()


@pytest.mark.parametrize("V,S,G,scale", [(3000, 64, 8, 1.0), (2000, 16, 5, 0.2), (1500, 96, 12, 1.0), (800, 40, 3, 3.0)])
def test_screening_pass_never_changes_a_result(V, S, G, scale):
    """the fp32 screening pass of the tau sweep against the all-fp64 sweep: identical haplotype trace, change
    counts, log-likelihoods and MAP record over a burn-in from a random state (many undecided steps, the screen suspends itself)
    and over a run from the generating state (nearly every step decided by the screen), and the counters that say so"""
    counts, tau_true, gamma_true = synth_counts(V, S, G, seed=11, depth_scale=scale)
    for start in ("random", "truth"):
        if start == "random":
            tau0, gamma0, eta0 = random_state(V, S, G, seed=12)
        else:
            tau0 = cbind.idx_to_onehot(tau_true)
            gamma0, eta0 = np.ascontiguousarray(gamma_true), 0.96 * np.eye(4) + 0.01
        out = []
        for on in (True, False):
            c = _lib.Context(0)
            c.set_counts(counts); c.set_state(tau0, gamma0, eta0); c.seed(77, ctr_seed=99)
            c.set_tau_screen(on)
            c.sweep_stats(reset=True)
            c.gibbs_update(25)
            tr = c.get_trace()
            taus = [c.get_tau_at(i) for i in (0, 7, 24)]
            out.append((tr, taus, c.get_star(), c.sweep_stats()))
            c.close()
        (tr_a, tau_a, star_a, st_a), (tr_b, tau_b, star_b, st_b) = out
        assert all(np.array_equal(x, y) for x, y in zip(tau_a, tau_b))
        for k in ("ll", "lp", "nchange", "gamma", "eta"):
            assert np.array_equal(tr_a[k], tr_b[k]), k
        assert star_a["lp"] == star_b["lp"] and np.array_equal(star_a["tau"], star_b["tau"])
        assert st_b[0] == st_b[1] and st_b[0] > 0                       # switched off: every wavefront-step counted as fp64
        assert st_a[0] == st_b[0] and st_a[1] <= st_a[0]
        if start == "truth" and scale >= 1.0:
            assert st_a[1] < 0.2 * st_a[0]                               # the screen decided most steps


@pytest.mark.parametrize("V,S,G", [(2902, 183, 2), (515, 96, 2), (303, 64, 4), (640, 48, 3), (700, 128, 8), (200, 300, 4)])
def test_factorize_tau_with_subnormal_start_values_and_tiny_abundances(ctx, V, S, G):
    """scripts/dbg/fuzz_nmft.py, round 5: 8e-4 of the components of a Dirichlet(0.01) draw (Init_NMFT.py:84) are below 1e-308, and with
    abundances that are as small in some samples R = tau . gamma is a SUBNORMAL number there; F / R is still finite in the reference
    (8.9e307 in the case found) while the hardware reciprocal of a subnormal is inf (rounds 2-4: NaN rows).  The first case is the one
    the fuzzer found; the others plant such rows: tau start values of 1e-308 / 1e-130, gamma columns of 1 / 1e-200."""
    counts, _, _ = synth_counts(V, S, min(G, 4), seed=V + S)
    tau0, gam0 = rn.nmft_random_initialize(np.random.RandomState(V + 3 * S + G), V, S, G)
    if V != 2902:
        gam0 = np.full((G, S), 1.0 / G)
        for s in (3, S - 1):                                       # two samples: one haplotype has it all, the others 1e-200
            gam0[:, s] = 1e-200
            gam0[s % G, s] = 1.0
        for v in range(0, V, 7):                                   # base 3 of every seventh position: a subnormal in one haplotype, 1e-130 in the others
            tau0[3 * V + v, :] = 1e-130
            tau0[3 * V + v, v % G] = 1e-308
    F = cbind.nmft_freq(counts)
    ctx.set_counts(counts)
    ctx.nmft_set(tau0, gam0)
    tc, gc = tau0.copy(), gam0.copy()
    n_ref, tr_ref = cbind.nmft_factorize_tau(F, tc, gc, max_iter=6, min_change=0.0)
    assert np.isfinite(tc).all() and np.isfinite(tr_ref[: n_ref + 1]).all()          # the reference's numbers are finite here
    n, tr = ctx.nmft_factorize(max_iter=6, min_change=0.0, fix_gamma=True)
    assert n == n_ref
    np.testing.assert_allclose(tr, tr_ref, rtol=1e-9)
    t, g = ctx.nmft_get()
    np.testing.assert_allclose(t, tc, rtol=1e-6, atol=1e-12)
