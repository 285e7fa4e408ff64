"""ctypes bindings to oracle/_build/liboracle.so (the C restatement).

TEST INFRASTRUCTURE ONLY -- see oracle/desman_oracle.c for the per-function
reference citations.  `build()` compiles the library with gcc if needed.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")


def build(force=False):
    if os.environ.get("DESMAN_ORACLE_SO"):          # e.g. the sanitizer build (`make -C oracle asan-test`)
        return os.environ["DESMAN_ORACLE_SO"]
    srcs = [os.path.join(_HERE, f) for f in ("desman_oracle.c", "stats_agg.c", "orc_log_table.h")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_mt_seed.argtypes = [C.c_void_p, C.c_ulong]
        L.orc_mt_u32.argtypes = [C.c_void_p]
        L.orc_mt_u32.restype = C.c_uint32
        L.orc_mt_fill_u32.argtypes = [C.c_void_p, _u32p, C.c_long]
        L.orc_setRNG.argtypes = [C.c_ulong]
        L.orc_sample_tau_u.argtypes = [_i64p, _f64p, _f64p, _i64p, C.c_int, C.c_int, C.c_int,
                                       _f64p, C.c_void_p]
        L.orc_sample_tau_u.restype = C.c_int
        L.orc_sample_tau.argtypes = [_i64p, _f64p, _f64p, _i64p, C.c_int, C.c_int, C.c_int]
        L.orc_sample_tau.restype = C.c_int
        L.orc_loglik.argtypes = [_u8p, _f64p, _f64p, _i64p, C.c_int, C.c_int, C.c_int]
        L.orc_loglik.restype = C.c_double
        L.orc_loglik_const.argtypes = [_i64p, C.c_int, C.c_int]
        L.orc_loglik_const.restype = C.c_double
        L.orc_logprior.argtypes = [_f64p, _f64p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double]
        L.orc_logprior.restype = C.c_double
        L.orc_logpost.argtypes = [_u8p, _f64p, _f64p, _i64p, C.c_int, C.c_int, C.c_int,
                                  C.c_double, C.c_double]
        L.orc_logpost.restype = C.c_double
        L.orc_nmft_freq.argtypes = [_i64p, C.c_int, C.c_int, _f64p]
        L.orc_nmft_objective.argtypes = [_f64p, _f64p, _f64p, C.c_int, C.c_int, C.c_int]
        L.orc_nmft_objective.restype = C.c_double
        L.orc_nmft_adjust.argtypes = [_f64p, _f64p, C.c_int, C.c_int, C.c_int]
        L.orc_nmft_update.argtypes = [_f64p, _f64p, _f64p, C.c_int, C.c_int, C.c_int]
        L.orc_nmft_update_tau.argtypes = [_f64p, _f64p, _f64p, C.c_int, C.c_int, C.c_int]
        L.orc_nmft_factorize.argtypes = [_f64p, _f64p, _f64p, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_double, C.c_void_p]
        L.orc_nmft_factorize.restype = C.c_int
        L.orc_nmft_factorize_tau.argtypes = L.orc_nmft_factorize.argtypes
        L.orc_nmft_factorize_tau.restype = C.c_int
        L.orc_nmft_get_tau.argtypes = [_f64p, C.c_int, C.c_int, _u8p]
        L.orc_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
        L.orc_stats_counter.argtypes = [_u8p, _f64p, _f64p, _i64p, C.c_int, C.c_int, C.c_int,
                                        C.c_uint64, C.c_uint32, _u64p, _u64p]
        L.orc_stats_expect.argtypes = [_u8p, _f64p, _f64p, _i64p, C.c_int, C.c_int, C.c_int,
                                       _f64p, _f64p, _f64p]
        L.orc_stats_agg.argtypes = [_u8p, _f64p, _f64p, _i64p, C.c_int, C.c_int, C.c_int,
                                    C.c_uint64, C.c_uint32, _u64p, _u64p, C.c_void_p, C.c_int]
        L.orc_stats_agg.restype = C.c_int
        L.orc_binom_test.argtypes = [C.c_int, C.c_uint32, C.c_double, C.c_double, C.c_uint64, C.c_int, _u32p, C.c_int]
        L.orc_mult4_test.argtypes = [C.c_uint32, _f64p, C.c_uint64, C.c_int, _u32p, C.c_int]
        L.orc_tlog.argtypes = [C.c_double]
        L.orc_tlog.restype = C.c_double
        L.orc_texp.argtypes = [C.c_double]
        L.orc_texp.restype = C.c_double
        L.orc_dirichlet_counter.argtypes = [_u64p, _u64p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                            C.c_uint64, C.c_uint32, _f64p, _f64p, _f64p]
        _lib = L
    return _lib


class MT19937:
    """GSL-flavoured MT19937 stream (seed 0 -> 4357; uniform = u32 / 2**32)."""

    def __init__(self, seed):
        self._buf = C.create_string_buffer(624 * 4 + 8)
        lib().orc_mt_seed(self._buf, int(seed) & 0xFFFFFFFFFFFFFFFF)

    def raw(self, n):
        out = np.empty(int(n), dtype=np.uint32)
        lib().orc_mt_fill_u32(self._buf, out, int(n))
        return out

    def uniform(self, n):
        return self.raw(n).astype(np.float64) / 4294967296.0


def onehot_to_idx(tau):
    """[V,G,4] one-hot -> [V,G] uint8 (first 1 wins, as c_sample_tau.c:116-122)."""
    return np.ascontiguousarray(np.argmax(np.asarray(tau) == 1, axis=2).astype(np.uint8))


def idx_to_onehot(idx):
    idx = np.asarray(idx)
    out = np.zeros(idx.shape + (4,), dtype=np.int64)
    np.put_along_axis(out, idx[..., None].astype(np.int64), 1, axis=2)
    return out


def sample_tau_u(tau, pi, eta, variants, u, want_logp=False):
    """In-place tau sweep with explicit uniforms; returns nchange[, logp]."""
    V, G, _ = tau.shape
    S = pi.shape[0]
    logp = np.empty((V, G, 4)) if want_logp else None
    n = lib().orc_sample_tau_u(tau, pi, eta, variants, V, G, S, np.ascontiguousarray(u, dtype=np.float64),
                               logp.ctypes.data if want_logp else None)
    return (n, logp) if want_logp else n


def initRNG():
    lib().orc_initRNG()


def setRNG(seed):
    lib().orc_setRNG(int(seed))


def freeRNG():
    lib().orc_freeRNG()


def sample_tau(tau, pi, eta, variants):
    V, G, _ = tau.shape
    return lib().orc_sample_tau(tau, pi, eta, variants, V, G, pi.shape[0])


def loglik(tau_idx, gamma, eta, variants):
    V, S, _ = variants.shape
    return lib().orc_loglik(tau_idx, gamma, eta, variants, V, tau_idx.shape[1], S)


def loglik_const(variants):
    return lib().orc_loglik_const(variants, variants.shape[0], variants.shape[1])


def logprior(gamma, eta, V, alpha=0.1, delta=0.1):
    S, G = gamma.shape
    return lib().orc_logprior(gamma, eta, V, G, S, alpha, delta)


def logpost(tau_idx, gamma, eta, variants, alpha=0.1, delta=0.1):
    V, S, _ = variants.shape
    return lib().orc_logpost(tau_idx, gamma, eta, variants, V, tau_idx.shape[1], S, alpha, delta)


def nmft_freq(variants):
    V, S, _ = variants.shape
    F = np.empty((4 * V, S))
    lib().orc_nmft_freq(variants, V, S, F)
    return F


def nmft_objective(F, tau, gam):
    G, S = gam.shape
    return lib().orc_nmft_objective(F, tau, gam, F.shape[0] // 4, G, S)


def nmft_adjust(tau, gam):
    G, S = gam.shape
    lib().orc_nmft_adjust(tau, gam, tau.shape[0] // 4, G, S)


def nmft_update(F, tau, gam):
    G, S = gam.shape
    lib().orc_nmft_update(F, tau, gam, F.shape[0] // 4, G, S)


def nmft_update_tau(F, tau, gam):
    G, S = gam.shape
    lib().orc_nmft_update_tau(F, tau, gam, F.shape[0] // 4, G, S)


def nmft_factorize(F, tau, gam, max_iter=5000, min_change=1e-5):
    G, S = gam.shape
    tr = np.full(max_iter + 1, np.nan)
    it = lib().orc_nmft_factorize(F, tau, gam, F.shape[0] // 4, G, S, max_iter, min_change, tr.ctypes.data)
    return it, tr[: it + 1]


def nmft_factorize_tau(F, tau, gam, max_iter=5000, min_change=1e-5):
    G, S = gam.shape
    tr = np.full(max_iter + 1, np.nan)
    it = lib().orc_nmft_factorize_tau(F, tau, gam, F.shape[0] // 4, G, S, max_iter, min_change,
                                      tr.ctypes.data)
    return it, tr[: it + 1]


def nmft_get_tau(tau, G):
    V = tau.shape[0] // 4
    out = np.empty((V, G), dtype=np.uint8)
    lib().orc_nmft_get_tau(tau, V, G, out)
    return out


def philox4x32_10(ctr, key):
    out = np.empty(4, dtype=np.uint32)
    lib().orc_philox4x32_10(np.asarray(ctr, dtype=np.uint32), np.asarray(key, dtype=np.uint32), out)
    return out


def stats_counter(tau_idx, gamma, eta, variants, seed, it):
    V, S, _ = variants.shape
    G = tau_idx.shape[1]
    mu = np.zeros((S, G), dtype=np.uint64)
    E = np.zeros((4, 4), dtype=np.uint64)
    lib().orc_stats_counter(tau_idx, gamma, eta, variants, V, G, S, int(seed), int(it), mu, E)
    return mu, E


def stats_expect(tau_idx, gamma, eta, variants):
    V, S, _ = variants.shape
    G = tau_idx.shape[1]
    e_mu = np.zeros((S, G)); v_mu = np.zeros((S, G)); e_E = np.zeros((4, 4))
    lib().orc_stats_expect(tau_idx, gamma, eta, variants, V, G, S, e_mu, v_mu, e_E)
    return e_mu, v_mu, e_E


def dirichlet_counter(sum_mu, esum, seed, it, alpha=0.1, delta=0.1, epsilon=1e-6):
    """(gamma [S,G], eta [4,4], rowprior [S+4]) of the counter-based Dirichlet draw specification."""
    S, G = sum_mu.shape
    g = np.empty((S, G)); e = np.empty((4, 4)); rp = np.empty(S + 4)
    lib().orc_dirichlet_counter(np.ascontiguousarray(sum_mu, dtype=np.uint64), np.ascontiguousarray(esum, dtype=np.uint64),
                                S, G, alpha, delta, epsilon, int(seed), int(it), g, e, rp)
    return g, e, rp


STATS_AGG = 2          # the aggregated specification the product runs by default (3 = the table exp / log variant, selectable)


def aggregate_over_patterns(tau_idx, variants):
    """spec 4's first step: positions with the same tau row (the same packed word) pool their counts in the LOWEST such position,
    every other position of the word is left without reads.  Cells (v, s), (v', s) of one word have the same weights for every
    observed base, and a sum of multinomials with one probability vector is one multinomial of the summed count: the mu/E sums keep
    their law (HaploSNP_Sampler.py:284-309 consumed as sums, :266,:276) while stage 1 draws once per word instead of once per position."""
    tau_idx = np.asarray(tau_idx)
    _, first, inv = np.unique(tau_idx, axis=0, return_index=True, return_inverse=True)
    rep = first[np.asarray(inv).reshape(-1)]                 # lowest position carrying each position's word
    out = np.zeros_like(variants)
    np.add.at(out, rep, variants)
    return out


def stats_agg(tau_idx, gamma, eta, variants, seed, it, want_ntab=False, spec=STATS_AGG):
    """the aggregated specification of the mu/E sums (oracle/stats_agg.c; spec 2 or 3): (sum_mu [S,G], Esum [4,4][, ntab [S,2^G]]).
    spec 4 = spec 2 on the counts pooled per tau word (aggregate_over_patterns); like the device it is spec 2 itself for G > 8."""
    V, S, _ = variants.shape
    G = tau_idx.shape[1]
    if spec == 4:
        if G <= 8:
            variants = np.ascontiguousarray(aggregate_over_patterns(tau_idx, variants))
        spec = 2
    mu = np.zeros((S, G), dtype=np.uint64)
    E = np.zeros((4, 4), dtype=np.uint64)
    nt = np.zeros((S, 1 << G), dtype=np.uint32) if want_ntab else None
    rc = lib().orc_stats_agg(tau_idx, gamma, eta, variants, V, G, S, int(seed), int(it), mu, E,
                             nt.ctypes.data if want_ntab else None, int(spec))
    if rc != 0:
        raise ValueError("orc_stats_agg: G outside 1..16 or spec not 2 / 3")
    return (mu, E, nt) if want_ntab else (mu, E)


def binom_test(kind, n, wa, wb, seed, nsamp, spec=STATS_AGG):
    out = np.empty(nsamp, dtype=np.uint32)
    lib().orc_binom_test(int(kind), int(n), float(wa), float(wb), int(seed), int(nsamp), out, int(spec))
    return out


def mult4_test(x, W, seed, nsamp, spec=STATS_AGG):
    out = np.empty((nsamp, 4), dtype=np.uint32)
    lib().orc_mult4_test(int(x), np.ascontiguousarray(W, dtype=np.float64), int(seed), int(nsamp), out, int(spec))
    return out


def tlog(x):
    return lib().orc_tlog(float(x))


def texp(y):
    return lib().orc_texp(float(y))
