"""Host mirror of desman/Init_NMFT.py: same class, constructor and method names,
the factorisation itself runs in the nmft_* HIP kernels behind the C ABI.

The random initial factors are drawn on the host from the caller's numpy
``RandomState`` in the reference's order (Init_NMFT.py:66-86), so they are
bit-identical to the reference's; everything after that is on the device.
"""
import logging

import numpy as np

from . import _lib


class Init_NMFT:
    """Initialises tau and gamma by tensor non-negative matrix factorisation (KL updates)."""

    BASE_PRIOR = 1.0

    def __init__(self, snps, rank, randomState, n_run=1, max_iter=5000, min_change=1.0e-5,
                 alpha_constant=0.01, device=0, ctx=None):
        snps = np.asarray(snps)
        self.V, self.S = snps.shape[0], snps.shape[1]
        self.G = int(rank)
        self.randomState = randomState
        self.n_run = n_run
        self.max_iter = max_iter
        self.min_change = min_change
        self.alpha = np.full(self.G, alpha_constant)
        self.alpha4 = np.full(4, alpha_constant)
        self.N = self.V * 4
        self._ctx = ctx if ctx is not None else _lib.Context(device)
        self._snps = np.ascontiguousarray(snps, dtype=np.int64)
        self._ctx.set_counts(self._snps)
        self._tau = np.zeros((self.N, self.G))
        self._gamma = np.zeros((self.G, self.S))
        self._dirty = True            # host factors newer than the device copy
        self.div_trace = None

    # tau / gamma are plain attributes in the reference (bin/desman:186 assigns .gamma);
    # assignments mark the device copy stale.
    @property
    def tau(self):
        return self._tau

    @tau.setter
    def tau(self, value):
        self._tau = np.ascontiguousarray(value, dtype=np.float64)
        self._dirty = True

    @property
    def gamma(self):
        return self._gamma

    @gamma.setter
    def gamma(self, value):
        self._gamma = np.ascontiguousarray(value, dtype=np.float64)
        self._dirty = True

    @property
    def freq_matrix(self):
        """F[v + a*V, s] = (x+1)/(n+4)  (Init_NMFT.py:49-60); host copy on demand."""
        x = self._snps.astype(np.float64) + self.BASE_PRIOR
        f = x / x.sum(axis=2)[:, :, None]
        return np.ascontiguousarray(np.transpose(f, (2, 0, 1)).reshape(self.N, self.S))

    def _push(self):
        if self._dirty:
            self._ctx.nmft_set(self._tau, self._gamma)
            self._dirty = False

    def _pull(self):
        self._tau, self._gamma = self._ctx.nmft_get()
        self._dirty = False

    # ---- initial draws (host, reference RNG order)
    def _draw_tau(self):
        # V*G successive dirichlet(alpha4) calls, v-major g-minor (:72-78);
        # dirichlet(a, size=n) consumes the stream exactly like n calls.
        d = self.randomState.dirichlet(self.alpha4, size=self.V * self.G).reshape(self.V, self.G, 4)
        return np.ascontiguousarray(np.transpose(d, (2, 0, 1)).reshape(self.N, self.G))

    def random_initialize(self):
        if self.G > 1:
            self.gamma = np.transpose(self.randomState.dirichlet(self.alpha, size=self.S))
        else:
            self.gamma = np.ones((self.G, self.S))
        self.tau = self._draw_tau()

    def random_initialize_tau(self):
        self.tau = self._draw_tau()

    # ---- factorisation (device)
    def _log_trace(self, trace):
        for it in range(0, len(trace) - 1, 100):
            logging.info('NTF Iter %d, div = %f' % (it, trace[it + 1]))

    def factorize(self):
        for _ in range(self.n_run):
            self.random_initialize()
            self._push()
            n, tr = self._ctx.nmft_factorize(self.max_iter, self.min_change, fix_gamma=False)
            self._pull()
            self.div_trace = tr
            self._log_trace(tr)

    @staticmethod
    def factorize_batch(objs):
        """factorize() of several objects of one shape at once (replicate chains): shared launches on the device, the
        stop test per chain (dsm_batch_nmft_factorize).  Same factors as factorize() on each."""
        objs = list(objs)
        if any(o.n_run != 1 for o in objs) or len({(o.max_iter, o.min_change) for o in objs}) != 1:
            raise ValueError("factorize_batch: n_run = 1 and equal max_iter / min_change expected")
        # a batch the kernels do not take (S > 128, G > 16, more than 8 chains) must leave every chain's numpy stream where
        # it was: the caller falls back to factorize(), which draws the same initial factors again
        states = [o.randomState.get_state() for o in objs]
        try:
            for o in objs:
                o.random_initialize()
                o._push()
            res = _lib.Context.batch_nmft_factorize([o._ctx for o in objs], objs[0].max_iter, objs[0].min_change, fix_gamma=False)
        except _lib.DesmanHipError:
            for o, st in zip(objs, states):
                o.randomState.set_state(st)
            raise
        for o, (n, tr) in zip(objs, res):
            o._pull()
            o.div_trace = tr
        return res

    @staticmethod
    def factorize_tau_batch(objs):
        """factorize_tau() (gamma fixed) of several objects of one shape at once; same factors as one by one"""
        objs = list(objs)
        if any(o.n_run != 1 for o in objs) or len({(o.max_iter, o.min_change) for o in objs}) != 1:
            raise ValueError("factorize_tau_batch: n_run = 1 and equal max_iter / min_change expected")
        states = [o.randomState.get_state() for o in objs]          # as in factorize_batch
        try:
            for o in objs:
                o.random_initialize_tau()
                o._push()
            res = _lib.Context.batch_nmft_factorize([o._ctx for o in objs], objs[0].max_iter, objs[0].min_change, fix_gamma=True)
        except _lib.DesmanHipError:
            for o, st in zip(objs, states):
                o.randomState.set_state(st)
            raise
        for o, (n, tr) in zip(objs, res):
            o._pull()
            o.div_trace = tr
        return res

    def factorize_tau(self):
        for _ in range(self.n_run):
            self.random_initialize_tau()
            self._push()
            n, tr = self._ctx.nmft_factorize(self.max_iter, self.min_change, fix_gamma=True)
            self._pull()
            self.div_trace = tr
            self._log_trace(tr)

    def div_objective(self):
        self._push()
        return self._ctx.nmft_objective()

    def get_gamma(self):
        return np.transpose(self._gamma)

    def get_tau(self):
        """argmax over the four bases -> one-hot int [V,G,4] (Init_NMFT.py:230-245)."""
        self._push()
        return self._ctx.nmft_get_tau()
