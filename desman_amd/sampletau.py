"""Drop-in replacement of DESMAN's Cython/GSL extension module ``sampletau``.

Same four callables, same argument meaning and the same exceptions as
sampletau/sampletau.pyx:21-57 in the reference; the work is done by the HIP
tau-sweep kernel behind the C ABI (include/desman_hip.h, legacy shim).  To use
it from reference-style code:  ``import desman_amd.sampletau as sampletau`` or
``sys.modules['sampletau'] = desman_amd.sampletau``.

There is no CPU fallback: a missing library or GPU raises DesmanHipError.
"""
import threading

import numpy as np

from . import _lib

# A thread that calls use_thread_local_rng() gets its own logical GSL stream (a host copy of the 625-word
# MT19937 state) instead of the process-global one of the C shim: several chains can then run
# concurrently in one process (desman_amd.chains), each continuing its own stream across its samplers.
_tls = threading.local()


def use_thread_local_rng(on=True, device=None):
    """`on`: this thread gets its own logical GSL stream AND its own device context for sample_tau (the C shim's
    context and stream are process-global: sampletau/c_sample_tau.c:24).  `device`: HIP device of that context
    (default: DESMAN_HIP_DEVICE or 0)."""
    ctx = getattr(_tls, "ctx", None)
    if ctx is not None:
        ctx.close()
    _tls.ctx = None
    _tls.enabled = bool(on)
    _tls.state = None
    if device is None:
        import os
        device = int(os.environ.get("DESMAN_HIP_DEVICE", "0"))
    _tls.device = int(device)


def _local():
    return getattr(_tls, "enabled", False)


def initRNG():
    """c_initRNG (sampletau.pyx:23-24): allocate the global MT19937 stream."""
    if _local():
        _tls.state = _lib.mt_seed_state(0)
        return
    _lib.check(_lib.load().dsm_initRNG())


def setRNG(seed):
    """c_setRNG (sampletau.pyx:28-29): `int seed` -> unsigned long."""
    if not isinstance(seed, (int, np.integer)):
        raise TypeError("an integer is required")
    if _local():
        if _tls.state is None:
            raise _lib.DesmanHipError("setRNG before initRNG")
        _tls.state = _lib.mt_seed_state(seed)
        return
    _lib.check(_lib.load().dsm_setRNG(int(seed) & 0xFFFFFFFFFFFFFFFF))


def freeRNG():
    """c_freeRNG (sampletau.pyx:33-34)."""
    if _local():
        _tls.state = None
        return
    _lib.check(_lib.load().dsm_freeRNG())


def _buf(a, name, dtype, ndim):
    # the Cython signature np.ndarray[T, ndim=N, mode="c"] not None (sampletau.pyx:38-41)
    if a is None:
        raise TypeError("Argument '%s' must not be None" % name)
    if not isinstance(a, np.ndarray):
        raise TypeError("Argument '%s' has incorrect type (expected numpy.ndarray, got %s)"
                        % (name, type(a).__name__))
    if a.dtype != dtype:
        raise ValueError("Buffer dtype mismatch, expected '%s' but got '%s'" % (np.dtype(dtype).name, a.dtype.name))
    if a.ndim != ndim:
        raise ValueError("Buffer has wrong number of dimensions (expected %d, got %d)" % (ndim, a.ndim))
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("ndarray is not C-contiguous")
    return a


def sample_tau(tau, pi, eta, variants):
    """sample_tau(tau, pi, eta, variants) -> nchange   (sampletau.pyx:38-57)

    tau [V,G,4] int64 one-hot (mutated in place), pi [S,G] f64, eta [4,4] f64,
    variants [V,S,4] int64.  Superset behaviour: shapes are cross-checked
    (the reference does not, sampletau.pyx:51-53)."""
    tau = _buf(tau, "tau", np.int64, 3)
    pi = _buf(pi, "pi", np.float64, 2)
    eta = _buf(eta, "eta", np.float64, 2)
    variants = _buf(variants, "variants", np.int64, 3)
    nV, nG = tau.shape[0], tau.shape[1]
    nS = pi.shape[0]
    if tau.shape[2] != 4 or pi.shape[1] != nG or eta.shape != (4, 4) or variants.shape != (nV, nS, 4):
        raise ValueError("inconsistent shapes: tau %s pi %s eta %s variants %s"
                         % (tau.shape, pi.shape, eta.shape, variants.shape))
    if _local():
        # this thread's own context and its own logical stream: the state is handed to the device for the sweep
        # and taken back afterwards, so reference-style code (Eta_Sampler.sampleTauC, HaploSNP_Sampler.update)
        # inside a desman_amd.chains worker thread draws from ITS stream, not from the process-global one
        if _tls.state is None:
            raise _lib.DesmanHipError("sample_tau: RNG not initialised (initRNG/setRNG)")
        if nV == 0:
            return 0
        if getattr(_tls, "ctx", None) is None:
            _tls.ctx = _lib.Context(getattr(_tls, "device", 0))
        ctx = _tls.ctx
        ctx.set_counts(variants)
        ctx.set_state(tau, pi, eta)
        ctx.set_tau_rng(_lib.RNG_MT19937)
        ctx.set_mt_state(_tls.state)
        n = ctx.sample_tau()
        tau[...] = ctx.get_state()[0]
        _tls.state = ctx.get_mt_state()
        return n
    rc = _lib.load().dsm_sample_tau(tau, pi, eta, variants, nV, nG, nS)
    if rc < 0:
        _lib.check(rc)
    return rc


def getRNGState():
    """(extension) the 625-word state of the global MT19937 stream."""
    if _local():
        if _tls.state is None:
            raise _lib.DesmanHipError("getRNGState: RNG not initialised")
        return _tls.state.copy()
    st = np.empty(625, dtype=np.uint32)
    _lib.check(_lib.load().dsm_getRNG_state(st))
    return st


def setRNGState(state):
    """(extension) restore the global MT19937 stream."""
    if _local():
        _tls.state = np.ascontiguousarray(state, dtype=np.uint32).copy()
        return
    _lib.check(_lib.load().dsm_setRNG_state(np.ascontiguousarray(state, dtype=np.uint32)))
