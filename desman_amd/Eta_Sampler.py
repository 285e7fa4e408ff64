"""Host mirror of desman/Eta_Sampler.py: which haplotypes carry which accessory gene.

Same constructor, attribute and method names as the reference class (GeneAssign.main drives it unchanged);
the data live on the GPU in ONE concatenated count tensor for all genes (desman_amd/csrc/genes.hip) instead of
a dict of per-gene arrays walked by Python loops.

Two samplers, chosen by ``rng``:

  "mt19937"  the reference's order of operations and of both random streams (the caller's numpy RandomState
             for the copy-number draws, the GSL MT19937 stream of ``sampletau`` for the tau sweeps).  One
             device step per (gene, haplotype); given the same seeds the copy numbers, tau and the MAP record
             are the reference's (tests/test_gpu_genes.py against tests/golden/gene_assign*.npz).
  "philox"   every gene advances together: per haplotype one sweep launch over all genes x both candidates and
             one draw launch, counter-based uniforms.  Same conditional distributions, different streams.

There is no CPU path: without the HIP library and a gfx950 device every method that computes raises.
"""
import logging

import numpy as np
import pandas as pd
from scipy.special import gammaln

from . import _lib, sampletau

MIN_DELTA = 1.0e-10
ETA_PENALTY = -1.0e3


class Eta_Sampler:

    def __init__(self, randomState, variants, covs, gamma, delta, cov_sd, epsilon, init_eta, max_iter=None,
                 tau_iter=None, max_eta=2, eta_scale=0.01, max_var=None, device=0, rng="mt19937", gene_base=0):
        if rng not in ("mt19937", "philox"):
            raise ValueError("rng must be 'mt19937' or 'philox'")
        self.randomState = randomState
        self.rng = rng
        # rng='philox': every host-side draw of gene c comes from a generator keyed by (one seed taken from
        # randomState, the gene's GLOBAL index, a phase counter) and every device draw is keyed by global gene /
        # indices and the row within the gene, so a run sharded over several samplers (gene_base = this shard's
        # first gene) gives the same result as one sampler over all genes
        self.gene_base = int(gene_base)
        self._phase = 0
        self._base_seed = int(randomState.randint(0, 2 ** 31 - 1)) if rng == "philox" else None
        self.delta = np.ascontiguousarray(np.transpose(delta), dtype=np.float64)        # [G,S]
        self.cov_sd = np.transpose(cov_sd)
        self.gamma = np.array(gamma, dtype=np.float64, order='C')
        self.cov = np.ascontiguousarray(covs.to_numpy(), dtype=np.float64)
        self.epsilon = np.array(epsilon, dtype=np.float64, order='C')
        self.S, self.G = self.gamma.shape
        self.C = self.cov.shape[0]
        self.ll = 0.0
        self.gene_ll = np.zeros(self.C)
        self.gene_llstar = np.zeros(self.C)
        self.genes = covs.index.tolist()
        self.gene_map = {gene: c for c, gene in enumerate(self.genes)}
        self.max_iter = 20 if max_iter is None else max_iter
        self.tau_iter = 5 if tau_iter is None else tau_iter
        self.max_eta = max_eta
        self.eta_scale = eta_scale

        # ---- per-gene variant rows (Eta_Sampler.py:72-105); optional random subsample of max_var rows
        rows_of = {}
        self._all_counts = np.zeros((0, self.S, 4), dtype=np.int64)
        if variants is not None and len(variants):
            rows_of = pd.Series(np.arange(len(variants))).groupby(variants.index.values, sort=False).indices
            table = variants.to_numpy()
            self._all_counts = np.ascontiguousarray(table.reshape(table.shape[0], table.shape[1] // 4, 4)).astype(np.int64)
        self._rows = {}
        self._rows_full = {}
        for gene in self.genes:
            rows = np.asarray(rows_of.get(gene, np.zeros(0, dtype=np.int64)), dtype=np.int64)
            if max_var is not None and len(rows) > max_var:
                self._rows_full[gene] = rows
                rows = rows[np.sort(self._gene_rs(self.gene_map[gene]).choice(len(rows), int(max_var), replace=False))]
            self._rows[gene] = rows
        self.gene_variants_full = {g: self._all_counts[r] for g, r in self._rows_full.items()}

        self.eta = np.array(init_eta, dtype=np.float64)
        self.eta[self.eta > self.max_eta - 1.0] = self.max_eta - 1.0
        self.eta_star = np.array(init_eta, dtype=np.float64)
        self.eta_store = np.zeros((self.max_iter, self.C, self.G))
        lp = np.arange(self.max_eta) * np.log(self.eta_scale)
        self.eta_log_prior = lp - np.log(np.sum(np.exp(lp)))

        self._dev = _lib.Genes(device)
        self._dev.set_gene_base(self.gene_base)
        self._upload()
        # ---- tau start of every gene with variants: NMFT with the masked gamma, then one sweep (:126-135)
        self._tau = np.zeros((self._Vtot, self.G, 4), dtype=np.int64)
        self._dev.set_state(self.eta.astype(np.int32), self._tau)
        if self._Vtot:
            start = self._draw_nmft_start(np.ones(self.C, dtype=bool))
            self._dev.nmft_tau(start, None)
            self._sweep_all(None)
            self._pull_tau()

    # ------------------------------------------------------------------ device plumbing
    def _upload(self):
        """(re)build the concatenated tensor from self._rows and push data + model to the device."""
        n_var = np.array([len(self._rows[g]) for g in self.genes], dtype=np.int64)
        self._gene_off = np.concatenate([[0], np.cumsum(n_var)]).astype(np.int32)
        self._Vtot = int(self._gene_off[-1])
        order = np.concatenate([self._rows[g] for g in self.genes]) if self._Vtot else np.zeros(0, dtype=np.int64)
        self._counts = self._all_counts[order] if self._Vtot else np.zeros((0, self.S, 4), dtype=np.int64)
        self.gene_V = {g: int(n) for g, n in zip(self.genes, n_var)}
        self.gene_variants = {g: (self._slice(self._counts, c) if n_var[c] else None) for c, g in enumerate(self.genes)}
        x = self._counts
        per_variant = (gammaln(x.sum(axis=2) + 1.0) - gammaln(x + 1.0).sum(axis=2)).sum(axis=1) if self._Vtot else np.zeros(0)
        self._v_const = per_variant                                                   # log_multinomial_pdf data part
        mult_const = np.array([per_variant[self._gene_off[c]:self._gene_off[c + 1]].sum() for c in range(self.C)])
        cov_const = -gammaln(self.cov + 1.0).sum(axis=1)
        self._dev.set_data(self._counts, self._gene_off, self.cov)
        self._consts = (cov_const, mult_const)
        self._dev.set_model(self.gamma, self.epsilon, self.delta, self.max_eta, self.eta_log_prior, cov_const, mult_const)

    def _slice(self, cat, c):
        return cat[self._gene_off[c]:self._gene_off[c + 1]]

    def _with_stream(self, fn):
        """run fn with the device MT19937 stream set to sampletau's global stream, and hand the stream back."""
        self._dev.set_mt_state(sampletau.getRNGState())
        try:
            return fn()
        finally:
            sampletau.setRNGState(self._dev.get_mt_state())

    def _gene_rs(self, c):
        """the RandomState behind gene c's host-side draws: the caller's (reference order) or the gene's own"""
        if self.rng != "philox":
            return self.randomState
        return np.random.RandomState([self._base_seed, self.gene_base + c, self._phase])

    def _sweep_all(self, mask, want_v_ll=False):
        """one masked sweep of every gene: GSL stream in gene order, or counter-based draws (rng='philox')"""
        if self.rng == "philox":
            return self._dev.sweep_all(mask, sweep=2, want_v_ll=want_v_ll)
        return self._with_stream(lambda: self._dev.sweep_all(mask, sweep=1, want_v_ll=want_v_ll))

    def _pull_tau(self):
        _, self._tau = self._dev.get_state(want_tau=True)

    @property
    def gene_tau(self):
        """per-gene view of the current tau ([V_c,G,4] int64), None for genes without variants"""
        return {g: (self._slice(self._tau, c) if self.gene_V[g] else None) for c, g in enumerate(self.genes)}

    def _draw_nmft_start(self, active):
        """Init_NMFT.random_initialize_tau for every active gene with variants, in gene order: V_c*G
        dirichlet(0.01 * 1_4) draws each (Init_NMFT.py:80-86) -> [Vtot,4,G]."""
        start = np.full((self._Vtot, 4, self.G), 0.25)
        for c, gene in enumerate(self.genes):
            V = self.gene_V[gene]
            if V and active[c]:
                d = self._gene_rs(c).dirichlet(np.full(4, 0.01), size=V * self.G).reshape(V, self.G, 4)
                start[self._gene_off[c]:self._gene_off[c + 1]] = np.transpose(d, (0, 2, 1))
        return start

    # ------------------------------------------------------------------ reference API
    def maskGamma(self, gamma, eta):
        keep = np.asarray(eta) != 0
        g = np.where(keep[None, :], gamma, 0.0)
        return g / g.sum(axis=1)[:, None]

    def logLikelihood(self):
        """sum of the per-gene terms: copy-number prior + Poisson coverage + multinomial variant counts."""
        self._dev.set_state(self.eta.astype(np.int32), None)
        self.gene_ll = self._dev.loglik()
        return float(np.sum(self.gene_ll))

    def sampleLogProb(self, adLogProbS):
        p = np.exp(adLogProbS - np.max(adLogProbS))
        p = p / np.sum(p, axis=0)
        return np.flatnonzero(self.randomState.multinomial(1, p, 1))[0]

    def sampleTauC(self, tau, variants, eta, gamma=None, epsilon=None):
        """reference signature: sweep ``tau`` (in place) of one gene's ``variants`` with gamma masked by eta"""
        gamma = self.gamma if gamma is None else gamma
        epsilon = self.epsilon if epsilon is None else epsilon
        return sampletau.sample_tau(tau, np.ascontiguousarray(self.maskGamma(gamma, eta)), np.ascontiguousarray(epsilon),
                                    np.ascontiguousarray(variants))

    def _coverage_terms(self, c, g):
        """sum_s log Poisson(cov | expected) for eta[c,g] = 0..max_eta-1 (update:243-259, incl. the clamp that
        log_Poisson applies in place to the eta[c,g] = 0 expectation before the others are built from it)"""
        rest = np.array(self.eta[c], copy=True)
        rest[g] = 0.0
        base = np.dot(rest, self.delta)
        base[base < MIN_DELTA] = MIN_DELTA
        lgam = gammaln(self.cov[c] + 1.0)
        out = np.empty(self.max_eta)
        for s in range(self.max_eta):
            ce = base + s * self.delta[g] if s else base
            ce = np.where(ce < MIN_DELTA, MIN_DELTA, ce)
            out[s] = (-lgam - ce + self.cov[c] * np.log(ce)).sum()
        return out

    def update(self):
        """max_iter Gibbs sweeps over all (gene, haplotype) copy numbers (Eta_Sampler.py:214-272)."""
        if self.rng == "philox":
            return self._update_batched()
        dev = self._dev
        dev.set_state(self.eta.astype(np.int32), None)
        self.ll = self.logLikelihood()
        self.eta_star = np.array(self.eta, copy=True)
        self.gene_llstar = np.array(self.gene_ll, copy=True)
        dev.set_mt_state(sampletau.getRNGState())
        try:
            for it in range(self.max_iter):
                for c, gene in enumerate(self.genes):
                    for g in range(self.G):
                        logvar, _ = dev.step_candidates(c, g)
                        lp = self.eta_log_prior + self._coverage_terms(c, g)
                        lp[0] += logvar[0]
                        lp[1:] += logvar[1]
                        pick = self.sampleLogProb(lp)
                        self.eta[c, g] = pick
                        dev.step_choose(c, g, pick)
                self.gene_ll = dev.loglik()
                self.ll = float(np.sum(self.gene_ll))
                logging.info('Gibbs Iter %d, nll = %f' % (it, self.ll))
                self.storeStarState(it)
                self.eta_store[it, ] = np.copy(self.eta)
        finally:
            sampletau.setRNGState(dev.get_mt_state())
        self._pull_tau()

    def _update_batched(self):
        dev = self._dev
        dev.set_state(self.eta.astype(np.int32), None)
        store, trace = dev.update(self.max_iter, reset_star=True)
        self.eta_store = store.astype(np.float64)
        for it in range(self.max_iter):
            logging.info('Gibbs Iter %d, nll = %f' % (it, trace[it].sum()))
        eta, self._tau = dev.get_state(want_tau=True)
        self.eta = eta.astype(np.float64)
        if self.max_iter:
            self.gene_ll = trace[-1].copy()
            self.ll = float(self.gene_ll.sum())
        star, self.gene_llstar = dev.get_star()
        self.eta_star = star.astype(np.float64)

    def storeStarState(self, iter):
        better = self.gene_ll > self.gene_llstar
        self.eta_star[better] = self.eta[better]
        self.gene_llstar[better] = self.gene_ll[better]

    def restoreFullVariants(self):
        """back to all variant rows of the genes that were subsampled to max_var; their tau restarts at zero"""
        if not self._rows_full:
            return
        old_off, old_tau = self._gene_off, self._tau
        restored = set(self._rows_full)
        for gene, rows in self._rows_full.items():
            self._rows[gene] = rows
        self._upload()
        tau = np.zeros((self._Vtot, self.G, 4), dtype=np.int64)
        for c, gene in enumerate(self.genes):
            if gene not in restored and self.gene_V[gene]:
                tau[self._gene_off[c]:self._gene_off[c + 1]] = old_tau[old_off[c]:old_off[c + 1]]
        self._tau = tau
        self._dev.set_state(self.eta.astype(np.int32), self._tau)

    def calcTauStar(self, eta, gamma=None, epsilon=None):
        """tau_iter sweeps with the copy numbers fixed to ``eta``; keeps, per variant, the tau with the best
        multinomial log-likelihood and every iteration's tau (Eta_Sampler.py:397-452).  A substitute ``gamma`` / ``epsilon`` is
        used where the reference uses it: by the sweeps and their likelihoods (:434-435), not by the NMF start, which takes the
        sampler's own gamma and eta (:420-423)."""
        eta = np.asarray(eta)
        sub = gamma is not None or epsilon is not None
        gamma_s = self.gamma if gamma is None else np.array(gamma, dtype=np.float64, order='C')
        eps_s = self.epsilon if epsilon is None else np.array(epsilon, dtype=np.float64, order='C')
        if gamma_s.shape != self.gamma.shape or eps_s.shape != (4, 4):
            raise ValueError("calcTauStar: gamma must be %s and epsilon (4, 4)" % (self.gamma.shape,))
        V = self._Vtot
        active = np.array([self.gene_V[g] > 0 and eta[c].sum() > 0 for c, g in enumerate(self.genes)])
        row_active = np.repeat(active, np.diff(self._gene_off)) if V else np.zeros(0, dtype=bool)
        ll_star = np.full(V, np.finfo(np.float64).min)
        tau_star = np.zeros((V, self.G, 4), dtype=np.int64)
        store = np.zeros((self.tau_iter, V, self.G, 4), dtype=np.int64)
        dev = self._dev
        self._phase += 1
        if V and active.any():
            start = self._draw_nmft_start(active)
            mask = np.where(active[:, None], self.eta, 0).astype(np.int32)          # the sampler's own eta (:421)
            n_it = dev.nmft_tau(start, mask)
            for c in np.flatnonzero(n_it >= 0):
                logging.info('Tau star NTF %d' % (c))
            self._pull_tau()
            tau_star[row_active] = self._tau[row_active]
        sweep_mask = np.where(active[:, None], eta, 0).astype(np.int32)
        if sub:                                                 # the sweeps and their likelihoods see the substitute model
            dev.set_model(gamma_s, eps_s, self.delta, self.max_eta, self.eta_log_prior, *self._consts)
        try:
            self._tau_star_sweeps(dev, V, active, row_active, sweep_mask, ll_star, tau_star, store)
        finally:
            if sub:
                dev.set_model(self.gamma, self.epsilon, self.delta, self.max_eta, self.eta_log_prior, *self._consts)
        self.gene_tau_star = {g: self._slice(tau_star, c) for c, g in enumerate(self.genes)}
        self.gene_ll_tau_star = {g: self._slice(ll_star, c) for c, g in enumerate(self.genes)}
        self.gene_tau_store = {g: store[:, self._gene_off[c]:self._gene_off[c + 1]] for c, g in enumerate(self.genes)}
        self._tau_star_cat, self._tau_store_cat = tau_star, store

    def _tau_star_sweeps(self, dev, V, active, row_active, sweep_mask, ll_star, tau_star, store):
        for it in range(self.tau_iter):
            total = 0.0
            if V and active.any():
                _, _, v_ll = self._sweep_all(sweep_mask, want_v_ll=True)
                self._pull_tau()
                ll = v_ll + self._v_const
                better = row_active & (ll > ll_star)
                ll_star[better] = ll[better]
                tau_star[better] = self._tau[better]
                store[it][row_active] = self._tau[row_active]
                for c in np.flatnonzero(active):
                    total += ll_star[self._gene_off[c]:self._gene_off[c + 1]].sum()
            logging.info('Tau star Iter %d, nll = %f' % (it, total))

    def getTauStar(self, variants):
        """(tau_star, tau_mean, positions, gene of every row) over all genes in gene order.  Like the reference
        the mean over the stored sweeps lands in an INTEGER array (Eta_Sampler.py:511,522): it is 1 only where
        every sweep agreed."""
        tau_star = np.array(self._tau_star_cat, copy=True)
        tau_mean = np.mean(self._tau_store_cat, axis=0).astype(np.int64) if self.tau_iter else np.zeros_like(tau_star)
        positions = np.zeros(self._Vtot, dtype=np.int64)
        contig_index = ["" for _ in range(self._Vtot)]
        pos_col = variants['Position'].to_numpy() if variants is not None else None
        for c, gene in enumerate(self.genes):
            lo, hi = self._gene_off[c], self._gene_off[c + 1]
            if hi > lo:
                contig_index[lo:hi] = [gene] * (hi - lo)
                positions[lo:hi] = pos_col[self._rows[gene]]
        return (tau_star, tau_mean, positions, contig_index)
