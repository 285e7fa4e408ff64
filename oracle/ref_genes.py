"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the accessory-gene path.

Restates desman/Eta_Sampler.py and GeneAssign.KLAssign as plain numpy + the C oracle's tau sweep
(oracle/desman_oracle.c: orc_sample_tau / orc_sample_tau_u).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product (desman_amd/) never does.

Pinned by tests/test_oracle_golden.py against tests/golden/gene_assign*.npz, which hold the states the
imported reference classes produced in this container (tests/golden/make_golden.py: gen_gene_assign).

Two drivers share the per-step arithmetic:
  * eta_update_reference_order -- Eta_Sampler.update (:214-272) exactly as the reference consumes its two
    RNG streams (numpy RandomState for the copy-number draws, the GSL MT19937 stream for the tau sweeps);
  * eta_update_batched -- the batched device sampler's specification: same conditionals, explicit uniforms.
"""
import numpy as np
from scipy.special import gammaln

from . import cbind

MIN_DELTA = 1.0e-10          # Eta_Sampler.py:18
ETA_PENALTY = -1.0e3         # Eta_Sampler.py:19


def eta_log_prior(max_eta=2, eta_scale=0.01):
    """Eta_Sampler.__init__ :139-145."""
    lp = np.arange(max_eta) * np.log(eta_scale)
    return lp - np.log(np.sum(np.exp(lp)))


def mask_gamma(gamma, eta_row):
    """maskGamma :147-157: drop the haplotypes without the gene, re-normalise every sample."""
    g = np.array(gamma, dtype=np.float64, copy=True)
    g[:, np.asarray(eta_row) == 0] = 0.0
    return g / g.sum(axis=1)[:, None]


def site_prob(tau, gamma_r, eps):
    """p[v,s,b] = sum_g gamma_r[s,g] eps[tau_vg, b]  (computeVarLLContrib :213, logLikelihoodGene :166)."""
    return np.einsum('vga,sg,ab->vsb', tau.astype(np.float64), gamma_r, eps)


def log_var(tau, gamma_r, eps, variants):
    """sum x log p  (computeVarLLContrib :214-215)."""
    return float((np.log(site_prob(tau, gamma_r, eps)) * variants).sum())


def poisson_sum(cov_row, cov_exp):
    """sum_s log_Poisson (:34-39); like the reference it clamps cov_exp IN PLACE."""
    cov_exp[cov_exp < MIN_DELTA] = MIN_DELTA
    return float((-gammaln(cov_row + 1.0) - cov_exp + cov_row * np.log(cov_exp)).sum())


def multinomial_const(variants):
    """data-only part of sum_{v,s} log_multinomial_pdf (:27-32)."""
    n = variants.sum(axis=2)
    return float((gammaln(n + 1.0) - gammaln(variants + 1.0).sum(axis=2)).sum())


def gene_loglik(eta_row, tau, variants, cov_row, gamma, eps, delta_gs, prior):
    """one gene's term of Eta_Sampler.logLikelihood (:182-202)."""
    ll = float(sum(prior[int(e)] for e in eta_row))
    ll += poisson_sum(cov_row, np.dot(np.asarray(eta_row, dtype=np.float64), delta_gs))
    if variants is not None and variants.shape[0] > 0:
        if np.sum(eta_row) > 0:
            ll += multinomial_const(variants) + log_var(tau, mask_gamma(gamma, eta_row), eps, variants)
        else:
            ll += variants.shape[0] * ETA_PENALTY
    return ll


def state_logprob(eta_row, g, cov_row, delta_gs, prior, lv0, lv1):
    """log-probabilities of eta[g] = 0..max_eta-1 given the two candidates' x log p (update :228-259)."""
    lp = np.array(prior, dtype=np.float64, copy=True)
    tmp = np.array(eta_row, dtype=np.float64, copy=True)
    tmp[g] = 0.0
    base = np.dot(tmp, delta_gs)
    lp[0] += poisson_sum(cov_row, base) + lv0                 # clamps `base` in place, as the reference does
    for s in range(1, len(lp)):
        lp[s] += poisson_sum(cov_row, base + s * delta_gs[g, :]) + lv1
    return lp


def _candidates(eta_row, g, tau, variants, gamma, eps, sweep):
    """the two computeVarLLContrib calls of one step; sweep(tau, gamma_r, k) -> new tau (one-hot copy)."""
    tmp = np.array(eta_row, copy=True)
    tmp[g] = 0
    V = 0 if variants is None else variants.shape[0]
    new0 = new1 = None
    lv0, lv1 = 0.0, 0.0
    if V > 0:
        if tmp.sum() > 0:
            g0 = mask_gamma(gamma, tmp)
            new0 = sweep(tau, g0, 0)
            lv0 = log_var(new0, g0, eps, variants)
        else:
            lv0 = -1.0e20
        tmp[g] = 1
        g1 = mask_gamma(gamma, tmp)
        new1 = sweep(tau, g1, 1)
        lv1 = log_var(new1, g1, eps, variants)
    return lv0, lv1, new0, new1


def sample_log_prob(rs, lp):
    """sampleLogProb :349-352 on a numpy RandomState."""
    d = np.exp(lp - np.max(lp))
    d = d / np.sum(d, axis=0)
    return int(np.flatnonzero(rs.multinomial(1, d, 1))[0])


def eta_update_reference_order(rs, eta, taus, variants, cov, gamma, eps, delta_gs, prior, n_iter, eta_star, llstar):
    """Eta_Sampler.update (:214-272).  eta [C,G] int (modified in place); taus / variants: per-gene lists
    (None / empty for genes without variants; taus modified in place).  The tau sweeps draw from the
    oracle's process-global GSL stream (cbind.setRNG).  Returns (eta_store, ll_trace, gene_ll)."""
    C, G = eta.shape

    def sweep(tau, gr, k):
        new = np.array(tau, copy=True)
        cbind.sample_tau(new, np.ascontiguousarray(gr), eps, variants_c)
        return new

    gene_ll = np.array([gene_loglik(eta[c], taus[c], variants[c], cov[c], gamma, eps, delta_gs, prior) for c in range(C)])
    eta_star[:] = eta
    llstar[:] = gene_ll
    store = np.zeros((n_iter, C, G))
    trace = np.zeros(n_iter)
    for it in range(n_iter):
        for c in range(C):
            variants_c = variants[c]
            for g in range(G):
                lv0, lv1, new0, new1 = _candidates(eta[c], g, taus[c], variants_c, gamma, eps, sweep)
                lp = state_logprob(eta[c], g, cov[c], delta_gs, prior, lv0, lv1)
                pick = sample_log_prob(rs, lp)
                eta[c, g] = pick
                if variants_c is not None and variants_c.shape[0] > 0:
                    taus[c] = new0 if pick == 0 else new1
        gene_ll = np.array([gene_loglik(eta[c], taus[c], variants[c], cov[c], gamma, eps, delta_gs, prior) for c in range(C)])
        trace[it] = gene_ll.sum()
        better = gene_ll > llstar
        eta_star[better] = eta[better]
        llstar[better] = gene_ll[better]
        store[it] = eta
    return store, trace, gene_ll


def eta_update_batched(eta, taus, variants, gene_off, cov, gamma, eps, delta_gs, prior, n_iter, u_tau, u_eta,
                       eta_star, llstar, reset_star=True):
    """Specification of dsm_genes_update with explicit uniforms: u_tau [n_iter][G][2][Vtot*G] raw 32-bit
    words (word of (gene row v, haplotype h) of candidate k at step g: [it][g][k][v*G+h]), u_eta
    [n_iter][C][G] uniforms of the inverse-CDF copy-number draw.  Same in-place conventions as above."""
    C, G = eta.shape
    gene_ll = np.array([gene_loglik(eta[c], taus[c], variants[c], cov[c], gamma, eps, delta_gs, prior) for c in range(C)])
    if reset_star:
        eta_star[:] = eta
        llstar[:] = gene_ll
    store = np.zeros((n_iter, C, G), dtype=np.int64)
    trace = np.zeros((n_iter, C))
    for it in range(n_iter):
        for g in range(G):
            for c in range(C):
                lo, hi = gene_off[c], gene_off[c + 1]

                def sweep(tau, gr, k):
                    u = u_tau[it, g, k, lo * G:hi * G].astype(np.float64) / 4294967296.0
                    new = np.array(tau, copy=True)
                    cbind.sample_tau_u(new, np.ascontiguousarray(gr), eps, variants[c], u)
                    return new

                lv0, lv1, new0, new1 = _candidates(eta[c], g, taus[c], variants[c], gamma, eps, sweep)
                lp = state_logprob(eta[c], g, cov[c], delta_gs, prior, lv0, lv1)
                ex = np.exp(lp - lp.max())
                us = u_eta[it, c, g] * ex.sum()
                pick, cum = len(lp) - 1, 0.0
                for s in range(len(lp) - 1):
                    cum += ex[s]
                    if us < cum:
                        pick = s
                        break
                eta[c, g] = pick
                if hi > lo:
                    taus[c] = new0 if pick == 0 else new1
        gene_ll = np.array([gene_loglik(eta[c], taus[c], variants[c], cov[c], gamma, eps, delta_gs, prior) for c in range(C)])
        trace[it] = gene_ll
        better = gene_ll > llstar
        eta_star[better] = eta[better]
        llstar[better] = gene_ll[better]
        store[it] = eta
    return store, trace


def gene_nmft_tau(rs, variants_c, gamma_r, G, max_iter=5000, min_change=1.0e-5):
    """Init_NMFT(gene).factorize_tau with gamma fixed to the masked gamma (Eta_Sampler.__init__ :128-134):
    draws the random start from `rs` like Init_NMFT.random_initialize_tau, returns (one-hot tau, updates)."""
    from . import ref_numpy as rn
    V = variants_c.shape[0]
    F = cbind.nmft_freq(variants_c)
    tau = rn.nmft_random_initialize_tau(rs, V, G)
    gam = np.ascontiguousarray(np.transpose(gamma_r))
    n, _ = cbind.nmft_factorize_tau(F, tau, gam, max_iter, min_change)
    return cbind.idx_to_onehot(cbind.nmft_get_tau(tau, G)), n


def kl_assign(cov, delta, eta, max_iter=10000, min_change=1.0e-4):
    """GeneAssign.KLAssign.factorize (GeneAssign.py:85-120) from given start values; returns
    (eta, updates, divergence)."""
    eps = np.finfo(np.float64).eps
    nzv = lambda a: np.where(a == 0, eps, a)
    dt = np.transpose(delta)                                    # [G,S]
    eta = np.array(eta, dtype=np.float64, copy=True)
    eta1 = dt.sum(1)[None, :]

    def objective():
        ca = eta @ dt
        return float((cov * np.log(nzv(cov) / nzv(ca)) - cov + ca).sum())

    divl, div, it = 0.0, objective(), 0
    while it < max_iter and abs(divl - div) > min_change:
        eta = eta * (nzv((nzv(cov) / nzv(eta @ dt)) @ dt.T) / nzv(eta1))
        eta = np.maximum(eta, eps)
        divl, div = div, objective()
        it += 1
    return eta, it, div


def calc_tau_star(rs, eta_self, eta_star, variants, gamma, eps, tau_iter, G, gamma_sub=None, eps_sub=None):
    """Eta_Sampler.calcTauStar (:397-452): NMFT start per gene (mask = the sampler's CURRENT eta and its own gamma, :420-421),
    tau_iter sweeps masked by eta_star, per-variant best tau under the full multinomial log-pdf.  ``gamma_sub`` / ``eps_sub``
    are the method's optional gamma / epsilon arguments: the sweeps and their likelihoods use them (:434-435), the start does not.
    Returns (tau_star list, ll_star list, tau_store list [tau_iter,V,G,4])."""
    gamma_sw = gamma if gamma_sub is None else gamma_sub
    eps_sw = eps if eps_sub is None else eps_sub
    C = len(variants)
    stars, lls, stores, taus = [], [], [], []
    for c in range(C):
        V = 0 if variants[c] is None else variants[c].shape[0]
        lls.append(np.full(V, np.finfo(np.float64).min))
        stars.append(np.zeros((V, G, 4), dtype=np.int64))
        stores.append(np.zeros((tau_iter, V, G, 4), dtype=np.int64))
        taus.append(np.zeros((V, G, 4), dtype=np.int64))
        if V > 0 and eta_star[c].sum() > 0:
            t, _ = gene_nmft_tau(rs, variants[c], mask_gamma(gamma, eta_self[c]), G)
            stars[c] = t.copy()
            taus[c] = t.copy()
    for it in range(tau_iter):
        for c in range(C):
            V = 0 if variants[c] is None else variants[c].shape[0]
            if V > 0 and eta_star[c].sum() > 0:
                gr = mask_gamma(gamma_sw, eta_star[c])
                cbind.sample_tau(taus[c], np.ascontiguousarray(gr), np.ascontiguousarray(eps_sw), variants[c])
                x = variants[c]
                n = x.sum(axis=2)
                ll = (gammaln(n + 1.0) - gammaln(x + 1.0).sum(axis=2) + (x * np.log(site_prob(taus[c], gr, eps_sw))).sum(axis=2)).sum(axis=1)
                better = ll > lls[c]
                lls[c][better] = ll[better]
                stars[c][better] = taus[c][better]
                stores[c][it] = taus[c]
    return stars, lls, stores
