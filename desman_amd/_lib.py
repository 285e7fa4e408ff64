"""ctypes binding of libdesman_hip.so -- the only native boundary of the package.

The product path has NO CPU fallback: if the HIP library is missing, cannot be
loaded, or no gfx950 device is visible, every compute entry point raises.
Signatures mirror include/desman_hip.h one to one.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DESMAN_HIP_LIB: another build of the same library (A/B builds of scripts/dbg: `make -C desman_amd/csrc EXTRA=... LIBNAME=...`)
LIB_PATH = os.environ.get("DESMAN_HIP_LIB") or os.path.join(_HERE, "lib", "libdesman_hip.so")
AB_LIB_PATH = os.path.join(_HERE, "lib", "libdesman_hip_ab.so")     # the experiment build (`make -C desman_amd/csrc ab`): A/B switches compiled in
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "desman_hip.h")

DSM_OK = 0
RNG_MT19937, RNG_PHILOX = 0, 1
STATS_AGG = 2        # version of the aggregated mu/E specification that runs by default (oracle/stats_agg.c); 3 = the table exp / log variant
K_NAMES = ("stats", "dirichlet", "tau", "finalize", "mt", "nmft_a", "nmft_gamma", "nmft_b", "stats2", "stats_big", "stats_pat")


class DesmanHipError(RuntimeError):
    pass


_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_vp, _i, _d = C.c_void_p, C.c_int, C.c_double

# name -> (restype, argtypes); every symbol include/desman_hip.h declares
SIGNATURES = {
    "dsm_last_error": (C.c_char_p, []),
    "dsm_device_count": (_i, []),
    "dsm_debug_ntab_probes": (_i, []),
    "dsm_version": (C.c_char_p, []),
    "dsm_initRNG": (_i, []),
    "dsm_setRNG": (_i, [C.c_ulong]),
    "dsm_freeRNG": (_i, []),
    "dsm_sample_tau": (_i, [_i64p, _f64p, _f64p, _i64p, _i, _i, _i]),
    "c_initRNG": (None, []),
    "c_setRNG": (None, [C.c_ulong]),
    "c_freeRNG": (None, []),
    "c_sample_tau": (_i, [_i64p, _f64p, _f64p, _i64p, _i, _i, _i]),
    "dsm_getRNG_state": (_i, [_u32p]),
    "dsm_setRNG_state": (_i, [_u32p]),
    "dsm_ctx_create": (_i, [C.POINTER(_vp), _i]),
    "dsm_ctx_destroy": (_i, [_vp]),
    "dsm_ctx_sync": (_i, [_vp]),
    "dsm_ctx_set_counts": (_i, [_vp, _i64p, _i, _i]),
    "dsm_ctx_set_state": (_i, [_vp, _i64p, _f64p, _f64p, _i]),
    "dsm_ctx_get_state": (_i, [_vp, _vp, _vp, _vp]),
    "dsm_ctx_set_gamma_eta": (_i, [_vp, _vp, _vp]),
    "dsm_ctx_set_priors": (_i, [_vp, _d, _d, _d]),
    "dsm_ctx_seed": (_i, [_vp, C.c_ulong, C.c_uint64]),
    "dsm_ctx_set_tau_rng": (_i, [_vp, _i]),
    "dsm_ctx_get_mt_state": (_i, [_vp, _u32p]),
    "dsm_ctx_set_mt_state": (_i, [_vp, _u32p]),
    "dsm_mt_seed_state": (_i, [C.c_ulong, _u32p]),
    "dsm_ctx_debug_mt_fill": (_i, [_vp, C.c_size_t, _u32p]),
    "dsm_ctx_sample_tau": (_i, [_vp, C.POINTER(_i), _vp]),
    "dsm_ctx_sample_stats": (_i, [_vp, C.c_uint32, _u64p, _u64p]),
    "dsm_ctx_stats_spec": (_i, [_vp]),
    "dsm_ctx_sweep_stats": (_i, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _i]),
    "dsm_ctx_set_tau_screen": (_i, [_vp, _i]),
    "dsm_ctx_set_tau_neartie": (_i, [_vp, _i]),
    "dsm_ctx_set_nmft_fused": (_i, [_vp, _i]),
    "dsm_ctx_set_nmft_persist": (_i, [_vp, _i]),
    "dsm_ctx_tau_launch_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "dsm_ctx_debug_log2f": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "dsm_debug_fdiv": (_i, [_i, _f64p, _f64p, _f64p, _i]),
    "dsm_release_device_caches": (_i, []),
    "dsm_ctx_force_stats_spec": (_i, [_vp, _i]),
    "dsm_ctx_debug_stage1": (_i, [_vp, C.c_uint32, _vp, _u64p]),
    "dsm_ctx_debug_binom": (_i, [_vp, _i, C.c_uint32, _f64p, C.c_uint64, _i, _u32p, _i]),
    "dsm_ctx_draw_gamma_eta": (_i, [_vp, C.c_uint32, _u64p, _u64p, _f64p, _f64p]),
    "dsm_ctx_loglik": (_i, [_vp, C.POINTER(_d), C.POINTER(_d)]),
    "dsm_ctx_gibbs_update": (_i, [_vp, _i]),
    "dsm_ctx_get_counters": (_i, [_vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "dsm_ctx_set_counters": (_i, [_vp, C.c_uint64, C.c_uint32]),
    "dsm_ctx_get_screen_state": (_i, [_vp, _u32p]),
    "dsm_ctx_set_screen_state": (_i, [_vp, _u32p]),
    "dsm_ctx_gibbs_update_sharded": (_i, [_vp, _i, _i, _i, _vp, _vp]),
    "dsm_ctx_gibbs_update_sharded_comm": (_i, [_vp, _i, _i, _i, _vp]),
    "dsm_comm_unique_id": (_i, [_vp]),
    "dsm_comm_create": (_i, [C.POINTER(_vp), _vp, _i, _i, _i]),
    "dsm_comm_destroy": (_i, [_vp]),
    "dsm_comm_rank": (_i, [_vp]),
    "dsm_comm_world": (_i, [_vp]),
    "dsm_comm_allgather_f64": (_i, [_vp, _f64p, _f64p, C.c_size_t]),
    "dsm_comm_allreduce_f64": (_i, [_vp, _vp, C.c_size_t, _i]),
    "dsm_comm_barrier": (_i, [_vp]),
    "dsm_device_read": (_i, [_i, _vp, _vp, C.c_size_t]),
    "dsm_device_write": (_i, [_i, _vp, _vp, C.c_size_t]),
    "dsm_batch_gibbs_update": (_i, [C.POINTER(_vp), _i, _i]),
    "dsm_batch_update_tau": (_i, [C.POINTER(_vp), _i, _i, C.POINTER(_vp), C.POINTER(_vp)]),
    "dsm_batch_nmft_factorize": (_i, [C.POINTER(_vp), _i, _i, _d, _i, C.POINTER(_i), _vp]),
    "dsm_ctx_update_tau": (_i, [_vp, _i, _f64p, _f64p]),
    "dsm_ctx_get_trace": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "dsm_ctx_get_star": (_i, [_vp, _vp, _vp, _vp, C.POINTER(_d), C.POINTER(_i)]),
    "dsm_ctx_get_tau_sum": (_i, [_vp, _i64p]),
    "dsm_ctx_get_tau_at": (_i, [_vp, _i, _i64p]),
    "dsm_nmft_set": (_i, [_vp, _f64p, _f64p, _i]),
    "dsm_nmft_get": (_i, [_vp, _vp, _vp]),
    "dsm_nmft_factorize": (_i, [_vp, _i, _d, _i, C.POINTER(_i), _vp]),
    "dsm_nmft_objective": (_i, [_vp, C.POINTER(_d)]),
    "dsm_nmft_get_tau": (_i, [_vp, _i64p]),
    "dsm_lrt_step": (_i, [_i, _f64p, _i32p, _i32p, _f64p, _d, _i, _i, _f64p, _f64p, _f64p]),
    "dsm_genes_create": (_i, [C.POINTER(_vp), _i]),
    "dsm_genes_destroy": (_i, [_vp]),
    "dsm_genes_set_data": (_i, [_vp, _vp, _i, _i, _i, _i32p, _f64p]),
    "dsm_genes_set_model": (_i, [_vp, _f64p, _f64p, _f64p, _i, _i, _f64p, _f64p, _f64p]),
    "dsm_genes_set_state": (_i, [_vp, _vp, _vp]),
    "dsm_genes_get_state": (_i, [_vp, _vp, _vp]),
    "dsm_genes_seed": (_i, [_vp, C.c_ulong, C.c_uint64]),
    "dsm_genes_set_gene_base": (_i, [_vp, _i]),
    "dsm_genes_get_mt_state": (_i, [_vp, _u32p]),
    "dsm_genes_set_mt_state": (_i, [_vp, _u32p]),
    "dsm_genes_nmft_tau": (_i, [_vp, _vp, _f64p, _i, _d, _vp]),
    "dsm_genes_sweep_all": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "dsm_genes_step_candidates": (_i, [_vp, _i, _i, _f64p, _vp]),
    "dsm_genes_step_choose": (_i, [_vp, _i, _i, _i]),
    "dsm_genes_update": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "dsm_genes_loglik": (_i, [_vp, _vp]),
    "dsm_genes_get_star": (_i, [_vp, _vp, _vp]),
    "dsm_genes_set_star": (_i, [_vp, _vp, _vp]),
    "dsm_kl_assign": (_i, [_i, _f64p, _f64p, _f64p, _i, _i, _i, _i, _d, C.POINTER(_i), C.POINTER(_d)]),
    "dsm_ctx_set_timing": (_i, [_vp, _i]),
    "dsm_ctx_get_timing": (_i, [_vp, _vp, _vp]),
    "dsm_kernel_name": (C.c_char_p, [_i]),
}

EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t)
_lib = None


def _digest(a):
    """128-bit digest of an array's bytes (xxh3 when the module is there: ~10 GB/s; blake2b otherwise)."""
    try:
        import xxhash
        return xxhash.xxh3_128_digest(memoryview(a).cast("B"))
    except ImportError:
        import hashlib
        return hashlib.blake2b(memoryview(a).cast("B"), digest_size=16).digest()


def load():
    """Load libdesman_hip.so (once).  Raises DesmanHipError if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DesmanHipError(
                "libdesman_hip.so not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C desman_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:
            raise DesmanHipError("cannot load %s: %s" % (LIB_PATH, e))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError = ABI drift: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != DSM_OK:
        raise DesmanHipError("libdesman_hip: error %d: %s" % (rc, load().dsm_last_error().decode()))
    return rc


def mt_seed_state(seed):
    """the 625-word MT19937 state gsl_rng_set(mt19937, seed) produces (position = 624)."""
    st = np.empty(625, dtype=np.uint32)
    check(load().dsm_mt_seed_state(int(seed) & 0xFFFFFFFFFFFFFFFF, st))
    return st


def release_device_caches():
    """frees the per-process, per-device caches of the library (MT19937 jump tables, placed subset tables); no context may be busy"""
    check(load().dsm_release_device_caches())


def debug_fdiv(kind, a, b):
    """test hook: a / b as the NMFT update divides (0 = fdiv_ext, 1 = fdiv_lo, 2 = fdiv; kernels_nmft.hip)"""
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
    b = np.ascontiguousarray(b, dtype=np.float64).reshape(-1)
    out = np.empty_like(a)
    check(load().dsm_debug_fdiv(int(kind), a, b, out, a.size))
    return out


def lrt_step(ffreq, maxA, maxB, eta, upperP, optimise, p, device=0):
    """one inner step of the likelihood-ratio variant filter on the GPU -> (p, MLL, BLL)."""
    ffreq = np.ascontiguousarray(ffreq, dtype=np.float64)
    V = ffreq.shape[0]
    p = np.ascontiguousarray(p, dtype=np.float64).copy()
    MLL = np.empty(V); BLL = np.empty(V)
    check(load().dsm_lrt_step(int(device), ffreq, np.ascontiguousarray(maxA, dtype=np.int32),
                              np.ascontiguousarray(maxB, dtype=np.int32), np.ascontiguousarray(eta, dtype=np.float64),
                              float(upperP), int(bool(optimise)), V, p, MLL, BLL))
    return p, MLL, BLL


def device_count():
    return load().dsm_device_count()


def ntab_probes():
    """how often this process has measured a subset table's place (include/desman_hip.h: dsm_debug_ntab_probes)"""
    return load().dsm_debug_ntab_probes()


def _ptr(a):
    return None if a is None else a.ctypes.data


class Context:
    """A device-resident chain context (dsm_ctx)."""

    def __init__(self, device=0):
        self._h = _vp()
        self.lib = load()
        check(self.lib.dsm_ctx_create(C.byref(self._h), int(device)))
        self.V = self.S = self.G = 0
        self.nG = 0
        self.n_trace = 0

    def close(self):
        if self._h:
            self.lib.dsm_ctx_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self.lib.dsm_ctx_sync(self._h))

    # ---- data / state
    def set_counts(self, variants):
        v = np.ascontiguousarray(variants, dtype=np.int64)
        if v.ndim != 3 or v.shape[2] != 4:
            raise ValueError("variants must be [V,S,4]")
        # re-uploading is skipped when the very same tensor is already resident (Init_NMFT and HaploSNP_Sampler of
        # one run share a context): same shape and same 128-bit digest of the bytes
        token = (v.shape, _digest(v))
        if getattr(self, "_counts_token", None) == token:
            return
        self._counts_token = None                        # a failed upload leaves no tensor resident
        check(self.lib.dsm_ctx_set_counts(self._h, v, v.shape[0], v.shape[1]))
        self.V, self.S = v.shape[0], v.shape[1]
        self.G = 0
        self._counts_token = token

    def set_state(self, tau, gamma, eta):
        tau = np.ascontiguousarray(tau, dtype=np.int64)
        gamma = np.ascontiguousarray(gamma, dtype=np.float64)
        eta = np.ascontiguousarray(eta, dtype=np.float64)
        if tau.shape != (self.V, gamma.shape[1], 4) or gamma.shape[0] != self.S or eta.shape != (4, 4):
            raise ValueError("state shapes do not match the count tensor")
        check(self.lib.dsm_ctx_set_state(self._h, tau, gamma, eta, gamma.shape[1]))
        self.G = gamma.shape[1]

    def get_state(self):
        tau = np.empty((self.V, self.G, 4), dtype=np.int64)
        gamma = np.empty((self.S, self.G))
        eta = np.empty((4, 4))
        check(self.lib.dsm_ctx_get_state(self._h, _ptr(tau), _ptr(gamma), _ptr(eta)))
        return tau, gamma, eta

    def set_gamma_eta(self, gamma=None, eta=None):
        g = None if gamma is None else np.ascontiguousarray(gamma, dtype=np.float64)
        e = None if eta is None else np.ascontiguousarray(eta, dtype=np.float64)
        check(self.lib.dsm_ctx_set_gamma_eta(self._h, _ptr(g), _ptr(e)))

    def set_priors(self, alpha=0.1, delta=0.1, epsilon=1e-6):
        check(self.lib.dsm_ctx_set_priors(self._h, alpha, delta, epsilon))

    def seed(self, mt_seed, ctr_seed=None):
        if ctr_seed is None:
            ctr_seed = (int(mt_seed) * 0x9E3779B97F4A7C15 + 0x243F6A8885A308D3) & 0xFFFFFFFFFFFFFFFF
        check(self.lib.dsm_ctx_seed(self._h, int(mt_seed) & 0xFFFFFFFFFFFFFFFF, int(ctr_seed)))

    def get_mt_state(self):
        st = np.empty(625, dtype=np.uint32)
        check(self.lib.dsm_ctx_get_mt_state(self._h, st))
        return st

    # ---- checkpoint / resume (the reference's own hook is dead code)
    def checkpoint(self):
        """everything that places the chain: state, MT19937 stream, counter-stream key and iteration counter (a dict of arrays:
        np.savez(path, **ctx.checkpoint()) writes it).  Counts, priors and the tau RNG mode are the caller's to keep."""
        tau, gamma, eta = self.get_state()
        key, it = self.counters()
        try:
            mt = self.get_mt_state()
        except DesmanHipError:                                   # never seeded: counter-based runs
            mt = np.zeros(0, dtype=np.uint32)
        return dict(tau=tau, gamma=gamma, eta=eta, mt_state=mt, ctr_seed=np.uint64(key), iter_ctr=np.uint32(it), screen=self.screen_state())

    def counters(self):
        """(key of the counter-based streams, iterations drawn so far) -- needs no resident state"""
        key, it = C.c_uint64(0), C.c_uint32(0)
        check(self.lib.dsm_ctx_get_counters(self._h, C.byref(key), C.byref(it)))
        return int(key.value), int(it.value)

    def screen_state(self):
        """the tau sweep's screening state: sweeps still to run without the fp32 screening pass, per launch parity"""
        out = np.zeros(2, dtype=np.uint32)
        check(self.lib.dsm_ctx_get_screen_state(self._h, out))
        return out

    def set_screen_state(self, words):
        check(self.lib.dsm_ctx_set_screen_state(self._h, np.ascontiguousarray(words, dtype=np.uint32)))

    def restore(self, ck):
        """put a chain saved by checkpoint() into this context (counts set already): it continues bit for bit"""
        self.set_state(np.ascontiguousarray(ck["tau"], dtype=np.int64), np.ascontiguousarray(ck["gamma"], dtype=np.float64),
                       np.ascontiguousarray(ck["eta"], dtype=np.float64))
        mt = np.asarray(ck["mt_state"], dtype=np.uint32)
        if mt.size == 625:
            self.set_mt_state(np.ascontiguousarray(mt))
        check(self.lib.dsm_ctx_set_counters(self._h, int(ck["ctr_seed"]), int(ck["iter_ctr"])))
        # the screening decisions of the saved chain (older checkpoints have none: the screen starts switched on, as in a fresh chain)
        self.set_screen_state(ck["screen"] if "screen" in ck else np.zeros(2, dtype=np.uint32))

    def set_mt_state(self, st):
        check(self.lib.dsm_ctx_set_mt_state(self._h, np.ascontiguousarray(st, dtype=np.uint32)))

    def debug_mt_fill(self, n):
        out = np.empty(int(n), dtype=np.uint32)
        check(self.lib.dsm_ctx_debug_mt_fill(self._h, int(n), out))
        return out

    def set_tau_rng(self, mode):
        check(self.lib.dsm_ctx_set_tau_rng(self._h, int(mode)))

    # ---- single steps
    def sample_tau(self, want_logp=False):
        n = _i(0)
        logp = np.empty((self.V, self.G, 4)) if want_logp else None
        check(self.lib.dsm_ctx_sample_tau(self._h, C.byref(n), _ptr(logp)))
        return (n.value, logp) if want_logp else n.value

    def sample_stats(self, it):
        mu = np.zeros((self.S, self.G), dtype=np.uint64)
        E = np.zeros((4, 4), dtype=np.uint64)
        check(self.lib.dsm_ctx_sample_stats(self._h, int(it), mu, E))
        return mu, E

    def debug_log2f(self, x):
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = np.empty_like(x)
        check(self.lib.dsm_ctx_debug_log2f(self._h, x.ctypes.data, out.ctypes.data, x.size))
        return out

    def set_tau_neartie(self, mode=-1):
        """which instantiation of the Gibbs loop's tau sweep runs: -1 = the one with the near-tie screen when the chain holds a haplotype
        that is rare in every sample (decided at the start of each gibbs_update call), 0 = never, 1 = always; same draws either way"""
        check(self.lib.dsm_ctx_set_tau_neartie(self._h, int(mode)))

    def set_tau_screen(self, on):
        """A/B switch: False = every step of the tau sweep in fp64 (same draws except in ~1e-13 near-ties)"""
        check(self.lib.dsm_ctx_set_tau_screen(self._h, 1 if on else 0))

    def tau_launch_info(self):
        """(workgroups a tau sweep launches, workgroups of that kernel resident on the device at once)"""
        a, b = _i(0), _i(0)
        check(self.lib.dsm_ctx_tau_launch_info(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_nmft_persist(self, mode=-1):
        """nmft_factorize as one persistent launch where the table fits: -1 / 1 = yes (default), 0 = the three-launch loop"""
        check(self.lib.dsm_ctx_set_nmft_persist(self._h, int(mode)))

    def set_nmft_fused(self, mode=-1):
        """gamma/control step of an NMFT update: -1 = by size, 0 = a launch of its own, 1 = one launch with the reduction, 3 = at the start of the update kernel (same results)"""
        check(self.lib.dsm_ctx_set_nmft_fused(self._h, int(mode)))

    def sweep_stats(self, reset=False):
        """(wavefront-steps of the tau sweeps so far, of which evaluated in fp64 because the screening pass
        could not decide them); reset=True also zeroes the totals"""
        a, b = C.c_uint64(0), C.c_uint64(0)
        check(self.lib.dsm_ctx_sweep_stats(self._h, C.byref(a), C.byref(b), 1 if reset else 0))
        return a.value, b.value

    def stats_spec(self):
        """2 / 3 = aggregated mu/E sampler (oracle/stats_agg.c: 2 = the default version, 3 = the table exp/log variant), 1 = per-read draws (orc_stats_counter)."""
        r = self.lib.dsm_ctx_stats_spec(self._h)
        if r < 0:
            check(r)
        return r

    def force_stats_spec(self, spec=0):
        """0 = shape rule, 1 = per-read draws everywhere, 2 (STATS_AGG) / 3 = that version of the aggregated sampler on small
        problems too (G <= 16), 4 = spec 2 over tau words (G <= 8; spec 2 where it does not apply)."""
        check(self.lib.dsm_ctx_force_stats_spec(self._h, int(spec)))

    def debug_stage1(self, it):
        nt = np.zeros((self.S, 1 << self.G), dtype=np.uint32)
        E = np.zeros((4, 4), dtype=np.uint64)
        check(self.lib.dsm_ctx_debug_stage1(self._h, int(it), nt.ctypes.data, E))
        return nt, E

    def debug_binom(self, kind, n, w, seed, nsamp, spec=STATS_AGG):
        w4 = np.zeros(4)
        w4[:len(w)] = w
        out = np.zeros((nsamp, 4) if kind == 2 else nsamp, dtype=np.uint32)
        check(self.lib.dsm_ctx_debug_binom(self._h, int(kind), int(n), w4, int(seed), int(nsamp), out.reshape(-1), int(spec)))
        return out

    def draw_gamma_eta(self, it, sum_mu, esum):
        g = np.empty((self.S, self.G)); e = np.empty((4, 4))
        check(self.lib.dsm_ctx_draw_gamma_eta(self._h, int(it), np.ascontiguousarray(sum_mu, dtype=np.uint64),
                                              np.ascontiguousarray(esum, dtype=np.uint64), g, e))
        return g, e

    def loglik(self):
        ll, lp = _d(0), _d(0)
        check(self.lib.dsm_ctx_loglik(self._h, C.byref(ll), C.byref(lp)))
        return ll.value, lp.value

    # ---- update loops
    def gibbs_update(self, n_iter):
        check(self.lib.dsm_ctx_gibbs_update(self._h, int(n_iter)))
        self.n_trace = int(n_iter)

    def gibbs_update_sharded(self, n_iter, v_offset, v_total, exchange):
        """n_iter iterations of ONE chain sharded by positions (include/desman_hip.h: dsm_ctx_gibbs_update_sharded): this context
        holds positions v_offset .. v_offset + V of v_total.  ``exchange(tab_ptr, n_tab, vec_ptr, n_vec)`` is the caller's
        all-reduce (sum) of n_tab uint32 at device address tab_ptr (0 words: skip) and n_vec float64 at vec_ptr, in place; an
        exception raised inside it aborts the call and is re-raised here."""
        err = []

        def cb(user, tab, n_tab, vec, n_vec):
            try:
                exchange(tab or 0, int(n_tab), vec, int(n_vec))
                return 0
            except BaseException as e:                       # noqa: BLE001 -- must not propagate through the C frame
                err.append(e)
                return 1
        fn = EXCHANGE_FN(cb)
        rc = self.lib.dsm_ctx_gibbs_update_sharded(self._h, int(n_iter), int(v_offset), int(v_total), C.cast(fn, _vp), None)
        if err:
            raise err[0]
        check(rc)
        self.n_trace = int(n_iter)

    def gibbs_update_sharded_comm(self, n_iter, v_offset, v_total, comm):
        """the same with the exchange done by the library over its own RCCL communicator (desman_amd/comm.py: Comm): the two
        all-reduces of an iteration are enqueued on the chain's stream, no host synchronisation, no torch"""
        check(self.lib.dsm_ctx_gibbs_update_sharded_comm(self._h, int(n_iter), int(v_offset), int(v_total), comm._h))
        self.n_trace = int(n_iter)

    @staticmethod
    def batch_gibbs_update(ctxs, n_iter):
        """n_iter Gibbs iterations of every chain in ``ctxs`` (contexts of one shape on one device, at most 8), one launch
        per kernel of the iteration for all of them (include/desman_hip.h: dsm_batch_gibbs_update)."""
        ctxs = list(ctxs)
        arr = (_vp * len(ctxs))(*[c._h.value for c in ctxs])
        check(load().dsm_batch_gibbs_update(arr, len(ctxs), int(n_iter)))
        for c in ctxs:
            c.n_trace = int(n_iter)

    def update_tau(self, gamma_store, eta_store):
        g = np.ascontiguousarray(gamma_store, dtype=np.float64)
        e = np.ascontiguousarray(eta_store, dtype=np.float64)
        check(self.lib.dsm_ctx_update_tau(self._h, g.shape[0], g, e))
        self.n_trace = g.shape[0]

    def get_trace(self):
        n = self.n_trace
        ll = np.empty(n); lp = np.empty(n); nch = np.empty(n, dtype=np.int32)
        g = np.empty((n, self.S, self.G)); e = np.empty((n, 4, 4))
        check(self.lib.dsm_ctx_get_trace(self._h, _ptr(ll), _ptr(lp), _ptr(nch), _ptr(g), _ptr(e)))
        return dict(ll=ll, lp=lp, nchange=nch, gamma=g, eta=e)

    def get_star(self):
        tau = np.empty((self.V, self.G, 4), dtype=np.int64)
        g = np.empty((self.S, self.G)); e = np.empty((4, 4))
        lp, it = _d(0), _i(0)
        check(self.lib.dsm_ctx_get_star(self._h, _ptr(tau), _ptr(g), _ptr(e), C.byref(lp), C.byref(it)))
        return dict(tau=tau, gamma=g, eta=e, lp=lp.value, it=it.value)

    def get_tau_sum(self):
        out = np.empty((self.V, self.G, 4), dtype=np.int64)
        check(self.lib.dsm_ctx_get_tau_sum(self._h, out))
        return out

    def get_tau_at(self, it):
        out = np.empty((self.V, self.G, 4), dtype=np.int64)
        check(self.lib.dsm_ctx_get_tau_at(self._h, int(it), out))
        return out

    # ---- NMFT
    def nmft_set(self, tau, gamma):
        tau = np.ascontiguousarray(tau, dtype=np.float64)
        gamma = np.ascontiguousarray(gamma, dtype=np.float64)
        G = gamma.shape[0]
        if tau.shape != (4 * self.V, G) or gamma.shape[1] != self.S:
            raise ValueError("NMFT factor shapes do not match the count tensor")
        check(self.lib.dsm_nmft_set(self._h, tau, gamma, G))
        self.nG = G

    def nmft_get(self):
        tau = np.empty((4 * self.V, self.nG)); gamma = np.empty((self.nG, self.S))
        check(self.lib.dsm_nmft_get(self._h, _ptr(tau), _ptr(gamma)))
        return tau, gamma

    def nmft_factorize(self, max_iter=5000, min_change=1e-5, fix_gamma=False):
        n = _i(0)
        tr = np.full(max_iter + 1, np.nan)
        check(self.lib.dsm_nmft_factorize(self._h, int(max_iter), float(min_change), int(bool(fix_gamma)),
                                          C.byref(n), _ptr(tr)))
        return n.value, tr[: n.value + 1]

    @staticmethod
    def batch_update_tau(ctxs, gamma_stores, eta_stores):
        """update_tau of every context in ``ctxs`` with shared launches (dsm_batch_update_tau); one (gamma_store, eta_store)
        pair of equal length per context"""
        ctxs = list(ctxs)
        gs = [np.ascontiguousarray(g, dtype=np.float64) for g in gamma_stores]
        es = [np.ascontiguousarray(e, dtype=np.float64) for e in eta_stores]
        n = gs[0].shape[0]
        if any(g.shape[0] != n for g in gs) or any(e.shape[0] != n for e in es):
            raise ValueError("batch_update_tau: traces of equal length expected")
        arr = (_vp * len(ctxs))(*[c._h.value for c in ctxs])
        gp = (_vp * len(ctxs))(*[g.ctypes.data for g in gs])
        ep = (_vp * len(ctxs))(*[e.ctypes.data for e in es])
        check(load().dsm_batch_update_tau(arr, len(ctxs), int(n), gp, ep))
        for c in ctxs:
            c.n_trace = int(n)

    @staticmethod
    def batch_nmft_factorize(ctxs, max_iter=5000, min_change=1e-5, fix_gamma=False):
        """nmft_factorize of every context in ``ctxs`` (one shape, one device, at most 8) with shared launches
        (include/desman_hip.h: dsm_batch_nmft_factorize) -> [(updates run, objective trace), ...]"""
        ctxs = list(ctxs)
        arr = (_vp * len(ctxs))(*[c._h.value for c in ctxs])
        n = (_i * len(ctxs))()
        tr = np.full((len(ctxs), max_iter + 1), np.nan)
        check(load().dsm_batch_nmft_factorize(arr, len(ctxs), int(max_iter), float(min_change), int(bool(fix_gamma)), n,
                                              tr.ctypes.data_as(_vp)))
        return [(int(n[k]), tr[k, : n[k] + 1].copy()) for k in range(len(ctxs))]

    def nmft_objective(self):
        d = _d(0)
        check(self.lib.dsm_nmft_objective(self._h, C.byref(d)))
        return d.value

    def nmft_get_tau(self):
        out = np.empty((self.V, self.nG, 4), dtype=np.int64)
        check(self.lib.dsm_nmft_get_tau(self._h, out))
        return out

    # ---- timing
    def set_timing(self, on):
        check(self.lib.dsm_ctx_set_timing(self._h, int(bool(on))))

    def get_timing(self):
        ms = np.zeros(len(K_NAMES)); n = np.zeros(len(K_NAMES), dtype=np.int64)
        check(self.lib.dsm_ctx_get_timing(self._h, _ptr(ms), _ptr(n)))
        return {k: (float(ms[i]), int(n[i])) for i, k in enumerate(K_NAMES)}


def kl_assign(cov, delta, eta, max_iter=10000, min_change=1.0e-4, device=0):
    """GeneAssign.KLAssign.factorize on the GPU: returns (eta [C,G], n_updates, divergence)."""
    cov = np.ascontiguousarray(cov, dtype=np.float64)
    delta = np.ascontiguousarray(delta, dtype=np.float64)
    eta = np.ascontiguousarray(eta, dtype=np.float64).copy()
    n, div = C.c_int(0), C.c_double(0.0)
    check(load().dsm_kl_assign(int(device), cov, delta, eta, cov.shape[0], cov.shape[1], delta.shape[1], int(max_iter),
                               float(min_change), C.byref(n), C.byref(div)))
    return eta, n.value, div.value


class Genes:
    """Device-resident accessory-gene sampler state (dsm_genes): all genes' variant rows in one tensor."""

    def __init__(self, device=0):
        self._h = _vp()
        check(load().dsm_genes_create(C.byref(self._h), int(device)))
        self.C = self.S = self.G = self.Vtot = 0

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            load().dsm_genes_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_data(self, variants, gene_off, cov):
        cov = np.ascontiguousarray(cov, dtype=np.float64)
        gene_off = np.ascontiguousarray(gene_off, dtype=np.int32)
        self.C, self.S = cov.shape
        self.Vtot = int(gene_off[-1])
        if self.Vtot:
            variants = np.ascontiguousarray(variants, dtype=np.int64)
            assert variants.shape == (self.Vtot, self.S, 4), variants.shape
        else:
            variants = None
        self.gene_off = gene_off
        check(load().dsm_genes_set_data(self._h, _ptr(variants), self.Vtot, self.S, self.C, gene_off, cov))

    def set_model(self, gamma, epsilon, delta_gs, max_eta, eta_log_prior, cov_const, mult_const):
        gamma = np.ascontiguousarray(gamma, dtype=np.float64)
        self.G = gamma.shape[1]
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        check(load().dsm_genes_set_model(self._h, gamma, f(epsilon), f(delta_gs), self.G, int(max_eta), f(eta_log_prior),
                                         f(cov_const), f(mult_const)))

    def set_state(self, eta=None, tau=None):
        eta = None if eta is None else np.ascontiguousarray(eta, dtype=np.int32)
        tau = None if tau is None else np.ascontiguousarray(tau, dtype=np.int64)
        check(load().dsm_genes_set_state(self._h, _ptr(eta), _ptr(tau)))

    def get_state(self, want_tau=True):
        eta = np.empty((self.C, self.G), dtype=np.int32)
        tau = np.zeros((self.Vtot, self.G, 4), dtype=np.int64) if want_tau else None
        check(load().dsm_genes_get_state(self._h, _ptr(eta), _ptr(tau)))
        return eta, tau

    def seed(self, mt_seed, ctr_seed=0x13198A2E03707344):
        check(load().dsm_genes_seed(self._h, int(mt_seed), int(ctr_seed)))

    def set_gene_base(self, gene_base):
        check(load().dsm_genes_set_gene_base(self._h, int(gene_base)))

    def get_mt_state(self):
        st = np.empty(625, dtype=np.uint32)
        check(load().dsm_genes_get_mt_state(self._h, st))
        return st

    def set_mt_state(self, st):
        check(load().dsm_genes_set_mt_state(self._h, np.ascontiguousarray(st, dtype=np.uint32)))

    def _mask(self, eta_mask):
        return None if eta_mask is None else np.ascontiguousarray(eta_mask, dtype=np.int32)

    def nmft_tau(self, tau_init, eta_mask=None, max_iter=5000, min_change=1.0e-5):
        """tau_init [Vtot,4,G]; returns the per-gene update counts (-1 = gene skipped)."""
        tau_init = np.ascontiguousarray(tau_init, dtype=np.float64)
        n = np.empty(self.C, dtype=np.int32)
        m = self._mask(eta_mask)
        check(load().dsm_genes_nmft_tau(self._h, _ptr(m), tau_init, int(max_iter), float(min_change), _ptr(n)))
        return n

    def sweep_all(self, eta_mask=None, sweep=True, want_v_ll=False):
        nch = np.empty(self.C, dtype=np.int32)
        lv = np.empty(self.C)
        vll = np.empty(self.Vtot) if want_v_ll else None
        m = self._mask(eta_mask)
        check(load().dsm_genes_sweep_all(self._h, _ptr(m), int(sweep), _ptr(nch), _ptr(lv), _ptr(vll)))
        return nch, lv, vll

    def step_candidates(self, c, g):
        lv = np.empty(2)
        sw = np.empty(2, dtype=np.int32)
        check(load().dsm_genes_step_candidates(self._h, int(c), int(g), lv, _ptr(sw)))
        return lv, sw

    def step_choose(self, c, g, value):
        check(load().dsm_genes_step_choose(self._h, int(c), int(g), int(value)))

    def update(self, n_iter, reset_star=True, u_tau_ext=None, u_eta_ext=None):
        store = np.empty((n_iter, self.C, self.G), dtype=np.int32)
        trace = np.empty((n_iter, self.C))
        ut = None if u_tau_ext is None else np.ascontiguousarray(u_tau_ext, dtype=np.uint32)
        ue = None if u_eta_ext is None else np.ascontiguousarray(u_eta_ext, dtype=np.float64)
        check(load().dsm_genes_update(self._h, int(n_iter), int(bool(reset_star)), _ptr(store), _ptr(trace), _ptr(ut), _ptr(ue)))
        return store, trace

    def loglik(self):
        ll = np.empty(self.C)
        check(load().dsm_genes_loglik(self._h, _ptr(ll)))
        return ll

    def get_star(self):
        es = np.empty((self.C, self.G), dtype=np.int32)
        ls = np.empty(self.C)
        check(load().dsm_genes_get_star(self._h, _ptr(es), _ptr(ls)))
        return es, ls

    def set_star(self, eta_star, gene_llstar):
        check(load().dsm_genes_set_star(self._h, _ptr(np.ascontiguousarray(eta_star, dtype=np.int32)),
                                        _ptr(np.ascontiguousarray(gene_llstar, dtype=np.float64))))
