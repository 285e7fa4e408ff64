"""Edge cases and the error channel of the C ABI (negative codes + message instead of the
reference's exit(1), c_sample_tau.c:200-203)."""
import numpy as np
import pytest

from desman_amd import _lib
from desman_amd.synth import synth_counts, random_state
from oracle import cbind

pytestmark = pytest.mark.gpu


def test_error_paths():
    ctx = _lib.Context(0)
    with pytest.raises(_lib.DesmanHipError, match="no count tensor"):
        ctx.lib.dsm_ctx_gibbs_update.restype  # noqa: B018  (keep the binding alive)
        _lib.check(ctx.lib.dsm_ctx_gibbs_update(ctx._h, 3))
    counts, _, _ = synth_counts(20, 4, 2, seed=1)
    ctx.set_counts(counts)
    with pytest.raises(_lib.DesmanHipError, match="no chain state"):
        _lib.check(ctx.lib.dsm_ctx_gibbs_update(ctx._h, 3))
    tau, gamma, eta = random_state(20, 4, 2, seed=1)
    ctx.set_state(tau, gamma, eta)
    ctx.set_tau_rng(_lib.RNG_MT19937)
    with pytest.raises(_lib.DesmanHipError, match="not seeded"):
        ctx.sample_tau()
    bad = counts.copy(); bad[0, 0, 0] = -1
    with pytest.raises(_lib.DesmanHipError, match="negative count"):
        ctx.set_counts(bad)
    bad = counts.copy(); bad[0, 0, 0] = 2 ** 31
    with pytest.raises(_lib.DesmanHipError, match="negative count or depth"):
        ctx.set_counts(bad)
    with pytest.raises(_lib.DesmanHipError, match="DSM_MAX_S"):
        ctx.set_counts(np.ones((2, 513, 4), dtype=np.int64))
    ctx.set_counts(counts)
    t33, g33, _ = random_state(20, 4, 33, seed=1)
    with pytest.raises(_lib.DesmanHipError, match="outside 1..32"):
        ctx.set_state(t33, g33, eta)
    with pytest.raises(_lib.DesmanHipError):
        _lib.Context(99)
    ctx.close()


def test_philox_tau_mode_is_deterministic_and_geometry_free():
    V, S, G = 500, 16, 4
    counts, _, _ = synth_counts(V, S, G, seed=2)
    tau, gamma, eta = random_state(V, S, G, seed=3)
    outs = []
    for _ in range(2):
        ctx = _lib.Context(0)
        ctx.set_counts(counts); ctx.set_state(tau, gamma, eta)
        ctx.seed(1, ctr_seed=777); ctx.set_tau_rng(_lib.RNG_PHILOX)
        ctx.gibbs_update(5)
        outs.append((ctx.get_state(), ctx.get_trace()))
        ctx.close()
    assert all(np.array_equal(a, b) for a, b in zip(outs[0][0], outs[1][0]))
    assert np.array_equal(outs[0][1]["ll"], outs[1][1]["ll"])          # bitwise reproducible run to run
    # the Philox uniforms are u32 / 2^32 of Philox(seed; v*G+g, iter, 'TAUU') -- check one sweep by hand
    ctx = _lib.Context(0)
    ctx.set_counts(counts); ctx.set_state(tau, gamma, eta)
    ctx.seed(1, ctr_seed=777); ctx.set_tau_rng(_lib.RNG_PHILOX)
    n = ctx.sample_tau()
    got, _, _ = ctx.get_state()
    key = [777, 0]
    u = np.array([cbind.philox4x32_10([i, 0, 0, 0x54415555], key)[0] for i in range(V * G)], dtype=np.float64) / 2 ** 32
    ref = tau.copy()
    assert cbind.sample_tau_u(ref, gamma, eta, counts, u) == n and np.array_equal(got, ref)
    ctx.close()


def test_max_sample_count_and_tiny_probabilities():
    V, S, G = 6, 512, 3                                           # DSM_MAX_S
    counts, _, _ = synth_counts(V, S, G, seed=4)
    tau, gamma, eta = random_state(V, S, G, seed=5)
    gamma[:, 0] = 2.220446049250313e-16                           # the NMFT clamp value (Init_NMFT.py:88-91)
    gamma /= gamma.sum(axis=1, keepdims=True)
    ctx = _lib.Context(0)
    ctx.set_counts(counts); ctx.set_state(tau, gamma, eta); ctx.seed(8)
    ref = tau.copy()
    n_ref = cbind.sample_tau_u(ref, gamma, eta, counts, cbind.MT19937(8).uniform(V * G))
    n = ctx.sample_tau()
    got, _, _ = ctx.get_state()
    assert n == n_ref and np.array_equal(got, ref)
    ll, _ = ctx.loglik()
    assert ll == pytest.approx(cbind.loglik(cbind.onehot_to_idx(got), gamma, eta, counts), rel=1e-12)
    # an exactly-zero mixture probability with a non-zero count: -inf/NaN propagate as in the reference
    eta0 = np.array([[1.0, 0, 0, 0], [0, 1.0, 0, 0], [0, 0, 1.0, 0], [0, 0, 0, 1.0]])
    ctx.set_state(tau, gamma, eta0)
    ll0, _ = ctx.loglik()
    assert not np.isfinite(ll0)
    ctx.close()


def test_contexts_release_their_device_memory():
    """create / use / destroy in a loop: the free device memory comes back (no leak in any of the objects)"""
    import ctypes
    from desman_amd import _lib
    from desman_amd.synth import synth_counts, random_state, synth_genes
    # the HIP runtime this process already uses (the one libdesman_hip.so is linked against): a second copy of the library, found by
    # name after another test pulled in a different one, has no device
    _lib.Context(0).close()
    paths = [ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln]
    rocm = [q for q in paths if q.startswith("/opt/rocm")]
    hip = ctypes.CDLL((rocm or paths or ["libamdhip64.so"])[0])

    def free_bytes():
        fr, tot = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipDeviceSynchronize() == 0 and hip.hipMemGetInfo(ctypes.byref(fr), ctypes.byref(tot)) == 0
        return fr.value

    from scipy.special import gammaln
    counts, _, _ = synth_counts(3000, 24, 5, seed=1)
    tau, gamma, eta = random_state(3000, 24, 5, seed=2)
    d = synth_genes(40, 12, 3, seed=3)
    off = np.concatenate([[0], np.cumsum(np.bincount(d['gene_of'], minlength=40))]).astype(np.int32)

    def cycle():
        ctx = _lib.Context(0)
        ctx.set_counts(counts)
        ctx.set_state(tau, gamma, eta)
        ctx.seed(1)
        ctx.gibbs_update(3)
        ctx.update_tau(np.repeat(gamma[None], 2, axis=0).copy(), np.repeat(eta[None], 2, axis=0).copy())
        ctx.close()
        gs = _lib.Genes(0)
        gs.set_data(d['counts'], off, d['cov'])
        lp = np.array([-0.01, -4.6])
        gs.set_model(d['gamma'], d['epsilon'], np.ascontiguousarray((d['gamma'] * d['total_mean'][:, None]).T), 2, lp,
                     np.zeros(40), np.zeros(40))
        gs.set_state(np.ones((40, 3), dtype=np.int32), np.zeros((d['counts'].shape[0], 3, 4), dtype=np.int64))
        gs.seed(1)
        gs.update(2)
        gs.close()

    cycle()
    free0 = free_bytes()
    for _ in range(25):
        cycle()
    free1 = free_bytes()
    assert free0 - free1 < 32 * 1024 * 1024, (free0, free1)


def test_mt19937_device_stream_is_gsl_stream_across_fills_and_handoffs():
    """the K-wide MT19937 generator (mt_fill_wide_kernel) against the oracle's serial gsl_rng_mt19937: raw words
    bit for bit over fills of awkward lengths (inside a block, exactly to a block edge, across the K = 1 / 2 / 4
    start-up, a million words), the standard (624 words + position) state after each fill, and a hand-off of that
    state to a second context"""
    ctx = _lib.Context(0)
    for seed in (0, 1, 4357, 2 ** 32 - 1):
        ctx.seed(seed)
        ref = cbind.MT19937(seed)
        for n in (1, 622, 1, 1, 624, 5, 619, 1248, 227, 2269, 3, 10 ** 6, 80000, 1):
            got = ctx.debug_mt_fill(n)
            assert np.array_equal(got, ref.raw(n)), (seed, n)
    # state hand-off: export, import into another context, both continue identically
    st = ctx.get_mt_state()
    assert 1 <= int(st[624]) <= 624
    c2 = _lib.Context(0)
    c2.set_mt_state(st)
    a, b = ctx.debug_mt_fill(5000), c2.debug_mt_fill(5000)
    c2.close()
    assert np.array_equal(a, b) and np.array_equal(a, ref.raw(5000))
    # a state produced by the host seeding routine (position 624: refill on first use)
    ctx.set_mt_state(_lib.mt_seed_state(99))
    assert np.array_equal(ctx.debug_mt_fill(3000), cbind.MT19937(99).raw(3000))
    ctx.close()


def test_mt19937_long_fills_from_several_compute_units_are_the_gsl_stream():
    """round 5: fills of 6 x 131 040 words and more run as chunks on several CUs, each from the state a GF(2) jump matrix makes of the
    stream's (kernels_gibbs.hip: mt_fill_parallel).  Raw words bit for bit against the oracle's serial gsl_rng_mt19937: 10^7 words
    (76.3 chunks = three rounds of 32 and a serial rest), lengths that end exactly on / one off a chunk and a round edge, from
    positions inside a block, the state left behind, short fills in between, a hand-off to another context."""
    D = 210 * 624
    ctx = _lib.Context(0)
    for seed in (5489, 0):
        ctx.seed(seed)
        ref = cbind.MT19937(seed)
        for n in (7, 6 * D, 1, 6 * D + 1, 600, 32 * D, 33 * D - 1, 10 ** 7, 3, 8 * D + 311):
            got = ctx.debug_mt_fill(n)
            want = ref.raw(n)
            assert np.array_equal(got, want), (seed, n, int(np.argmax(got != want)))
    st = ctx.get_mt_state()
    c2 = _lib.Context(0)
    c2.set_mt_state(st)
    a, b = ctx.debug_mt_fill(7 * D + 5), c2.debug_mt_fill(7 * D + 5)
    c2.close()
    assert np.array_equal(a, b) and np.array_equal(a, ref.raw(7 * D + 5))
    ctx.close()


def test_mt19937_stream_position_across_chunked_calls():
    """the sweeps' words are generated in chunks of 1, 2, 3, 4, 6, 8, 8, ... sweeps ahead of their reader, into two slots
    (api.hip: SweepWords): every sweep of a long updateTau gets exactly its V*G words of the GSL stream (bit-identical
    haplotypes over 30 sweeps = through the ramp and several reuses of both slots), and the stream position after any mix
    of calls is the reference's"""
    V, S, G, n = 150, 8, 3, 30
    counts, _, _ = synth_counts(V, S, G, seed=90)
    tau0, gamma0, eta0 = random_state(V, S, G, seed=91)
    rng = np.random.default_rng(2)
    gs = np.ascontiguousarray(rng.dirichlet(np.ones(G), size=(n, S)))
    es = np.ascontiguousarray(np.broadcast_to(eta0, (n, 4, 4)))
    ctx = _lib.Context(0)
    ctx.set_counts(counts); ctx.set_state(tau0, gamma0, eta0); ctx.seed(4242)
    ctx.update_tau(gs, es)
    mt = cbind.MT19937(4242)
    ref = tau0.copy()
    for it in range(n):
        cbind.sample_tau_u(ref, gs[it], es[it], counts, mt.uniform(V * G))
        assert np.array_equal(ctx.get_tau_at(it), ref), it
    # ... a Gibbs call (9 sweeps), one single sweep, and the next raw words
    ctx.gibbs_update(9)
    ctx.sample_tau()
    mt.raw(10 * V * G)
    assert np.array_equal(ctx.debug_mt_fill(1000), mt.raw(1000))
    ctx.close()


def test_hardware_log2_error_bound_of_the_screening_pass():
    """the screening pass of the tau sweep budgets the hardware v_log_f32 at a few ulp; measured here
    over the whole range it is used on (mixture values in [1e-30, 1]) and around 1, where log2 changes sign"""
    ctx = _lib.Context(0)
    rng = np.random.default_rng(5)
    x = np.concatenate([np.exp(rng.uniform(np.log(1e-30), 0.0, 2_000_000)), 1.0 - rng.uniform(0, 1e-3, 200_000),
                        rng.uniform(0.5, 1.0, 500_000), np.array([1.0, 0.5, 0.25, 1e-30])]).astype(np.float32)
    got = ctx.debug_log2f(x).astype(np.float64)
    ref = np.log2(x.astype(np.float64))
    err = np.abs(got - ref)
    ulp = np.spacing(np.abs(ref).astype(np.float32)).astype(np.float64)
    # absolute error: <= 2 ulp of the result away from 1, <= 2^-22 where the result is tiny
    assert np.all(err <= np.maximum(2.0 * ulp, 2.0 ** -22)), float((err / np.maximum(ulp, 2.0 ** -23)).max())
    assert got[-4] == 0.0 and got[-3] == -1.0 and got[-2] == -2.0
    ctx.close()


def test_the_place_of_the_subset_table_changes_no_number(tmp_path):
    """The mu/E pass's subset table starts where its memory-side atomics cost least -- measured per chain at run time,
    so the place differs from process to process.  A chain run with the measured place, with the allocator's place and with two fixed
    places gives the same traces, sums and final state (the switches are read from the environment once: child processes)."""
    import os
    import subprocess
    import sys
    code = r'''
import sys; sys.path.insert(0, %r)
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts, random_state
V, S, G = 700, 64, 6
counts, _, _ = synth_counts(V, S, G, seed=21)
tau, gam, eta = random_state(V, S, G, seed=5)
c = _lib.Context(0); c.set_counts(counts); c.seed(3); c.set_state(tau, gam, eta); c.force_stats_spec(_lib.STATS_AGG)
c.gibbs_update(12)
tr = c.get_trace(); t, g, e = c.get_state()
mu, es = c.sample_stats(77)
np.savez(sys.argv[1], ll=tr["ll"], lp=tr["lp"], nch=tr["nchange"], t=t, g=g, e=e, mu=mu, es=es)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    # DESMAN_HIP_NTAB_OFF (a place given from outside) exists only in the experiment build (-DDSM_AB_SWITCHES)
    ab = {"DESMAN_HIP_LIB": _lib.AB_LIB_PATH}
    for env_extra in ({}, {"DESMAN_HIP_NTAB_TUNE": "0"}, dict(ab, DESMAN_HIP_NTAB_OFF="768"), dict(ab, DESMAN_HIP_NTAB_OFF="4096")):
        path = str(tmp_path / ("o%d.npz" % len(outs)))
        r = subprocess.run([sys.executable, "-c", code, path], env=dict(os.environ, **env_extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(path))
    for o in outs[1:]:
        for k in outs[0].files:
            assert np.array_equal(outs[0][k], o[k]), k


def test_the_place_of_the_subset_table_is_not_measured_any_more():
    """Rounds 2-5 timed stage 1 at eight places of every new subset table and kept the fastest (what the table's memory-side atomics cost
    depended on its physical address: 44 vs 54-61 us at config 3), round 5 pooled the measured tables.  Round 6's row map puts the four
    64 B lines of a subset's row into four different rows (kernels_stats.hip: stats_ntab_swz): every place costs the same, 11 % less than
    the best one did (profiles/r06_swz_scan.txt), so the product library measures nothing -- and a chain is the same chain whichever
    table it gets, alone or next to another."""
    from desman_amd import _lib
    from desman_amd.synth import synth_counts, random_state
    V, S, G = 900, 64, 7
    counts, _, _ = synth_counts(V, S, G, seed=31)
    tau, gam, eta = random_state(V, S, G, seed=6)
    finals = []
    p0 = _lib.ntab_probes()
    for k in range(3):
        c = _lib.Context(0); c.set_counts(counts); c.seed(9); c.set_state(tau, gam, eta); c.force_stats_spec(_lib.STATS_AGG)
        c.gibbs_update(6)
        finals.append((c.get_trace()["ll"].copy(), c.get_state()[0].copy()))
        c.close()
    a = _lib.Context(0); a.set_counts(counts); a.seed(9); a.set_state(tau, gam, eta); a.force_stats_spec(_lib.STATS_AGG); a.gibbs_update(6)
    b = _lib.Context(0); b.set_counts(counts); b.seed(9); b.set_state(tau, gam, eta); b.force_stats_spec(_lib.STATS_AGG); b.gibbs_update(6)
    finals.append((a.get_trace()["ll"].copy(), a.get_state()[0].copy()))
    finals.append((b.get_trace()["ll"].copy(), b.get_state()[0].copy()))
    a.close(); b.close()
    assert _lib.ntab_probes() == p0
    for ll, t in finals[1:]:
        assert np.array_equal(ll, finals[0][0]) and np.array_equal(t, finals[0][1])


def test_release_device_caches_frees_and_everything_is_rebuilt_on_demand():
    """dsm_release_device_caches (100 MB of jump tables and up to 32 placed subset tables stay for the life of the process): after
    it a long MT19937 fill rebuilds the jump tables and a chain allocates its table afresh -- same words, same sums."""
    from desman_amd.synth import synth_counts, random_state
    V, S, G = 400, 64, 4
    counts, _, _ = synth_counts(V, S, G, seed=3)
    tau, gamma, eta = random_state(V, S, G, seed=4)

    def run():
        c = _lib.Context(0)
        try:
            c.set_counts(counts); c.set_state(tau, gamma, eta); c.seed(5, ctr_seed=6)
            c.force_stats_spec(2)
            mu, E = c.sample_stats(1)
            w = c.debug_mt_fill(7 * 131040 + 17)                      # long enough for the parallel generator
            return mu, E, w
        finally:
            c.close()

    a = run()
    b = run()
    _lib.release_device_caches()
    c_ = run()
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    for x, y in zip(a, c_):
        assert np.array_equal(x, y)
