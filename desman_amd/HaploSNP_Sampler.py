"""Host mirror of desman/HaploSNP_Sampler.py for the path `bin/desman` drives.

Same class name, constructor signature, attributes and method names; the Gibbs
iterations run entirely on the device (dsm_ctx_gibbs_update).  V-sized arrays
are fetched from HBM only when asked for (tau_store is materialised lazily).
Not mirrored (never reached by the CLI): the pure-Python
samplers, Chib marginal likelihood, DIC, assignTau, the 4^G tauStates table.
"""
import logging

import numpy as np

from . import _lib
from . import sampletau as _sampletau


class HaploSNP_Sampler:

    def __init__(self, snps, G, randomState, fixed_tau=None, burn_iter=None, max_iter=None,
                 alpha_constant=0.1, delta_constant=0.1, epsilon=1.0e-6, device=0, ctx=None):
        self.burn_iter = 250 if burn_iter is None else burn_iter
        self.max_iter = 250 if max_iter is None else max_iter
        self.randomState = randomState
        self.G = int(G)
        snps = np.asarray(snps)
        self.V, self.S = snps.shape[0], snps.shape[1]
        self.variants = np.array(snps, dtype=np.int64, order='C')
        self.epsilon = epsilon
        self.delta_constant = delta_constant
        self.delta = np.full(4, delta_constant)
        self.alpha_constant = alpha_constant
        self.alpha = np.full(self.G, alpha_constant)
        # the constructor consumes the caller's RNG exactly like the reference (:63, :72)
        self.gamma = self.randomState.dirichlet(self.alpha, size=self.S)
        if fixed_tau is None:
            tri = self.randomState.randint(0, 4, self.V * self.G).reshape(self.V, self.G)
            self.tau = np.zeros((self.V, self.G, 4), dtype=np.int64)
            np.put_along_axis(self.tau, tri[..., None], 1, axis=2)
        else:
            self.tau = np.reshape(fixed_tau, (self.V, self.G, 4)).astype(np.int64)
        self.tauIndices = np.zeros(self.V, dtype=np.int64)
        self.eta = 0.96 * np.identity(4) + 0.01 * np.ones((4, 4))
        self._alloc_stores()
        self.ll = 0.0
        self.lp = 0.0
        self._ctx = ctx if ctx is not None else _lib.Context(device)
        self._ctx.set_counts(self.variants)
        self._ctx.set_priors(alpha_constant, delta_constant, epsilon)
        self._tau_sum = None
        self._have_trace = False
        self._keyed = False
        self.mt_state = None            # a GSL stream of the sampler's own (update_batch: several chains in one thread);
                                        # None = the stream of the sampletau module, as in the reference

    def _alloc_stores(self):
        self.gamma_store = np.zeros((self.max_iter, self.S, self.G))
        self.eta_store = np.zeros((self.max_iter, 4, 4))
        self.ll_store = np.zeros(self.max_iter)
        self.lp_store = np.zeros(self.max_iter)
        self.nchange_store = np.zeros(self.max_iter, dtype=np.int64)

    def calcK(self):
        return self.V * self.G + self.S * (self.G - 1)

    # ---- RNG plumbing: the tau uniforms continue the process-global GSL-compatible stream
    def _bind_rng(self):
        st = self.mt_state if self.mt_state is not None else _sampletau.getRNGState()   # raises if initRNG()/setRNG() were not called
        if not self._keyed:
            # key the counter-based streams (mu/E, gamma, eta) once per sampler object
            # a function of the stream position only: deterministic whatever else runs in the process
            key = ((int(st[0]) << 32) ^ int(st[1]) ^ (int(st[2]) << 16) ^ (0x9E3779B97F4A7C15 * (int(st[624]) + 1))) \
                & 0xFFFFFFFFFFFFFFFF
            self._ctx.seed(1, ctr_seed=key)
            self._keyed = True
        self._ctx.set_mt_state(st)

    def _release_rng(self):
        if self.mt_state is not None:
            self.mt_state = self._ctx.get_mt_state()
        else:
            _sampletau.setRNGState(self._ctx.get_mt_state())

    def _push_state(self):
        self._ctx.set_state(np.ascontiguousarray(self.tau, dtype=np.int64),
                            np.ascontiguousarray(self.gamma, dtype=np.float64),
                            np.ascontiguousarray(self.eta, dtype=np.float64))
        self.G = self._ctx.G

    # ---- checkpoint / resume (optional; the reference's hook, Output_Results.output_Pickled_haploSNP, is dead code)
    def save_checkpoint(self, path):
        """the chain between two update() calls: state, both stream positions, the key of the counter-based streams.  A sampler
        built on the same table that loads it continues bit for bit (tests/test_gpu_host.py)."""
        self._bind_rng()                                      # keys the counter streams if no update() has run yet
        self._release_rng()
        key, it = self._ctx.counters()                        # (no resident state needed: also valid before the first update())
        mt = self.mt_state if self.mt_state is not None else _sampletau.getRNGState()
        np.savez(path, tau=np.asarray(self.tau, dtype=np.int64), gamma=self.gamma, eta=self.eta, mt_state=np.asarray(mt, dtype=np.uint32),
                 ctr_seed=np.uint64(key), iter_ctr=np.uint32(it), G=self.G, own_stream=self.mt_state is not None,
                 screen=self._ctx.screen_state())

    def load_checkpoint(self, path):
        z = np.load(path)
        if int(z["G"]) != self.G or z["tau"].shape != (self.V, self.G, 4) or z["gamma"].shape != (self.S, self.G):
            raise ValueError("checkpoint of another shape: tau %s, gamma %s" % (z["tau"].shape, z["gamma"].shape))
        self.tau, self.gamma, self.eta = z["tau"].copy(), z["gamma"].copy(), z["eta"].copy()
        self.updateTauIndices()
        if bool(z["own_stream"]):
            self.mt_state = z["mt_state"].copy()
        else:
            _sampletau.setRNGState(z["mt_state"])             # the module's stream, as in the run that saved it
        ck = dict(tau=self.tau, gamma=self.gamma, eta=self.eta, mt_state=z["mt_state"], ctr_seed=z["ctr_seed"], iter_ctr=z["iter_ctr"])
        if "screen" in z.files:
            ck["screen"] = z["screen"]
        self._ctx.restore(ck)
        self._keyed = True                                    # the counter streams keep their key and position

    # ---- the Gibbs loop
    def update(self):
        """max_iter Gibbs iterations (HaploSNP_Sampler.py:334-365), on the device."""
        self._push_state()
        self._bind_rng()
        self._ctx.gibbs_update(self.max_iter)
        self._release_rng()
        self._collect(prefix='nlp')

    @staticmethod
    def update_batch(samplers, on_chain=None):
        """update() of several chains of one shape on one device at once: one kernel launch per step of the iteration for
        all of them (dsm_batch_gibbs_update; the replicate chains of a G value, scripts/runDesman.sh:15-21).  Every sampler
        needs a GSL stream of its own (``mt_state``).  The mu/E pass of a batch is the aggregated sampler."""
        samplers = list(samplers)
        n = samplers[0].max_iter
        if any(s.max_iter != n for s in samplers) or any(s.mt_state is None for s in samplers):
            raise ValueError("update_batch: samplers need equal max_iter and a GSL stream each (mt_state)")
        for s in samplers:
            s._push_state()
            s._bind_rng()
        _lib.Context.batch_gibbs_update([s._ctx for s in samplers], n)
        for s in samplers:
            if on_chain is not None:
                on_chain(s)
            s._release_rng()
            s._collect(prefix='nlp')

    @staticmethod
    def updateTau_batch(samplers, on_chain=None):
        """updateTau() of several chains of one shape at once (dsm_batch_update_tau); a GSL stream each (mt_state)"""
        samplers = list(samplers)
        n = samplers[0].max_iter
        if any(s.max_iter != n for s in samplers) or any(s.mt_state is None for s in samplers):
            raise ValueError("updateTau_batch: samplers need equal max_iter and a GSL stream each (mt_state)")
        for s in samplers:
            s._push_state()
            s._bind_rng()
        _lib.Context.batch_update_tau([s._ctx for s in samplers], [s.gamma_store[:n] for s in samplers],
                                      [s.eta_store[:n] for s in samplers])
        for s in samplers:
            if on_chain is not None:
                on_chain(s)
            s._release_rng()
            gs, es = s.gamma_store, s.eta_store
            s._collect(prefix='nll', keep_gamma_eta=True)
            s.gamma_store, s.eta_store = gs, es

    def updateTau(self):
        """tau-only sweeps driven by gamma_store / eta_store (HaploSNP_Sampler.py:383-407)."""
        self._push_state()
        self._bind_rng()
        self._ctx.update_tau(self.gamma_store[:self.max_iter], self.eta_store[:self.max_iter])
        self._release_rng()
        gs, es = self.gamma_store, self.eta_store
        self._collect(prefix='nll', keep_gamma_eta=True)
        self.gamma_store, self.eta_store = gs, es

    def _collect(self, prefix, keep_gamma_eta=False):
        tr = self._ctx.get_trace()
        n = self.max_iter
        self.ll_store[:n] = tr["ll"]
        self.lp_store[:n] = tr["lp"]
        self.nchange_store[:n] = tr["nchange"]
        if not keep_gamma_eta:
            self.gamma_store[:n] = tr["gamma"]
            self.eta_store[:n] = tr["eta"]
        tau, gamma, eta = self._ctx.get_state()
        self.tau = tau
        if not keep_gamma_eta:
            self.gamma, self.eta = gamma, eta
        star = self._ctx.get_star()
        self.tau_star, self.lp_star, self.iter_star = star["tau"], star["lp"], star["it"]
        if not keep_gamma_eta:
            self.gamma_star, self.eta_star = star["gamma"], star["eta"]
        if n > 0:
            self.ll, self.lp = float(tr["ll"][-1]), float(tr["lp"][-1])
        for it in range(0, n, 10):
            logging.info('Gibbs Iter %d, no. changed = %d, %s = %f' % (it, tr["nchange"][it], prefix, tr["lp"][it]))
        self._tau_sum = None
        self._have_trace = True
        self.updateTauIndices()

    # ---- deterministic functions
    def logLikelihood(self, cGamma, cTau, cEta):
        self._ctx.set_state(np.ascontiguousarray(cTau, dtype=np.int64), np.ascontiguousarray(cGamma, dtype=np.float64),
                            np.ascontiguousarray(cEta, dtype=np.float64))
        return self._ctx.loglik()[0]

    def logPosterior(self, cGamma, cTau, cEta):
        self._ctx.set_state(np.ascontiguousarray(cTau, dtype=np.int64), np.ascontiguousarray(cGamma, dtype=np.float64),
                            np.ascontiguousarray(cEta, dtype=np.float64))
        return self._ctx.loglik()[1]

    def mapTauState(self, tauState):
        G = tauState.shape[0]
        w = 4 ** (G - 1 - np.arange(G, dtype=object))
        return int((np.argmax(tauState, axis=1).astype(object) * w).sum())

    def updateTauIndices(self):
        # base-4 index of the haplotype pattern (HaploSNP_Sampler.py:224-231); exact for G <= 31
        idx = np.argmax(self.tau, axis=2).astype(np.int64)
        if self.G <= 31:
            w = 4 ** (self.G - 1 - np.arange(self.G, dtype=np.int64))
            self.tauIndices = (idx * w[None, :]).sum(axis=1)
        else:
            self.tauIndices = np.array([self.mapTauState(self.tau[v]) for v in range(self.V)], dtype=object)

    def calculateSND(self, tau):
        """pairwise single-nucleotide differences between haplotypes (:712-730)."""
        idx = np.argmax(tau, axis=2)
        return (idx[:, :, None] != idx[:, None, :]).sum(axis=0)

    def compSND(self, tau1, tau2):
        i1, i2 = np.argmax(tau1, axis=2), np.argmax(tau2, axis=2)
        return (i1[:, :, None] != i2[:, None, :]).sum(axis=0)

    def variableTau(self, tau):
        idx = np.argmax(tau, axis=2)
        return (idx != idx[:, :1]).any(axis=1)

    def removeDegenerate(self):
        """merge haplotypes that are identical at every position (:771-832): the lower
        index survives, gamma columns are summed, survivors keep their order."""
        snd = self.calculateSND(self.tau)
        deleted = np.zeros(self.G, dtype=bool)
        absorbed = [[] for _ in range(self.G)]
        for g in range(self.G):
            for h in range(g + 1, self.G):
                if not deleted[h] and snd[g, h] == 0:
                    deleted[h] = True
                    absorbed[g].append(h)
        keep = [g for g in range(self.G) if not deleted[g]]
        gamma_new = np.zeros((self.S, len(keep)))
        for k, g in enumerate(keep):
            gamma_new[:, k] = self.gamma[:, g]
            for h in absorbed[g]:
                gamma_new[:, k] += self.gamma[:, h]
        self.tau = np.ascontiguousarray(self.tau[:, keep, :])
        self.gamma = gamma_new
        self.G = len(keep)
        self.alpha = np.full(self.G, self.alpha_constant)
        self.gamma_store = np.zeros((self.max_iter, self.S, self.G))
        self._have_trace = False
        self._tau_sum = None
        self.updateTauIndices()

    # ---- posterior summaries of the last update() (HaploSNP_Sampler.py:463-483,834-839)
    def meanDeviance(self):
        return -2.0 * np.mean(self.ll_store)

    def gammaMean(self):
        return np.mean(self.gamma_store, axis=0)

    def etaMean(self):
        return np.mean(self.eta_store, axis=0)

    def _tausum(self):
        if self._tau_sum is None:
            if not self._have_trace:
                raise _lib.DesmanHipError("no update() has run: tau_store is empty")
            self._tau_sum = self._ctx.get_tau_sum()
        return self._tau_sum

    def tauMean(self):
        return self._tausum() / float(self.max_iter)

    def probabilisticTau(self):
        return self._tausum() / float(self.max_iter)

    @property
    def tau_store(self):
        """[max_iter, V, G, 4] int64, fetched from the device trace on demand."""
        out = np.empty((self.max_iter, self.V, self.G, 4), dtype=np.int64)
        for it in range(self.max_iter):
            out[it] = self._ctx.get_tau_at(it)
        return out
