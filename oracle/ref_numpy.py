"""numpy restatement of the Python-level half of DESMAN's hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

These functions follow the reference's *serial* semantics, including the order
in which numpy's legacy ``RandomState`` stream is consumed, so that -- fed the
same ``RandomState`` -- they reproduce the reference bit for bit.  They use
pure-Python loops where the reference does and are meant for small cases only.
Each function cites the reference lines it follows (paths relative to the
upstream DESMAN tree).  Pinned against the imported reference by
tests/golden/make_golden.py -> tests/golden/*.npz.
"""
import math

import numpy as np
from scipy.special import gammaln

from . import cbind

EPS = np.finfo(np.float64).eps


# --------------------------------------------------------------------------
# HaploSNP_Sampler
# --------------------------------------------------------------------------
def sampler_ctor_draws(rs, V, S, G, alpha=0.1):
    """RNG consumed by HaploSNP_Sampler.__init__ (HaploSNP_Sampler.py:63,72-76):
    returns (gamma0 [S,G], tau0 one-hot [V,G,4])."""
    gamma = rs.dirichlet(np.full(G, alpha), size=S)
    tri = rs.randint(0, 4, V * G).reshape(V, G)
    return gamma, cbind.idx_to_onehot(tri)


def sample_mu(rs, tau, gamma, eta, variants):
    """sampleMu (HaploSNP_Sampler.py:284-309).  Returns (E [V,S,4,4] as
    [v,s,observed,true], mu [V,S,4,G])."""
    V, S, _ = variants.shape
    G = gamma.shape[1]
    T = np.einsum('ijk,lj,km->ilmkj', tau, gamma, eta)       # [v,s,obs,true,g]
    Tg = T.sum(axis=4)
    E = np.zeros((V, S, 4, 4), dtype=np.int64)
    mu = np.zeros((V, S, 4, G), dtype=np.int64)
    for v in range(V):
        for s in range(S):
            tm = Tg[v, s]
            tm = tm / tm.sum(axis=1)[:, None]
            for a in range(4):
                E[v, s, a, :] = rs.multinomial(variants[v, s, a], tm[a, :])
                for b in range(4):
                    if E[v, s, a, b] > 0:
                        w = T[v, s, a, b, :]
                        w = w / w.sum()
                        mu[v, s, a, :] += rs.multinomial(E[v, s, a, b], w)
    return E, mu


def sample_gamma(rs, mu, alpha=0.1, epsilon=1e-6):
    """sampleGamma (HaploSNP_Sampler.py:263-273)."""
    sum_mu = mu.sum(axis=(0, 2))
    S, G = sum_mu.shape
    gamma = np.empty((S, G))
    for s in range(S):
        gamma[s, :] = rs.dirichlet(alpha + sum_mu[s, :])
    return clamp_renorm_gamma(gamma, epsilon)


def clamp_renorm_gamma(gamma, epsilon=1e-6):
    """the deterministic tail of sampleGamma (:271-273)."""
    gamma = gamma.copy()
    gamma[gamma < epsilon] = epsilon
    return gamma / gamma.sum(axis=1)[:, None]


def sample_eta(rs, E, delta=0.1):
    """sampleEta (HaploSNP_Sampler.py:275-281)."""
    Esum = E.sum(axis=(0, 1))
    eta = np.empty((4, 4))
    for a in range(4):
        eta[a, :] = rs.dirichlet(delta + Esum[:, a])
    return eta


def log_likelihood(tau, gamma, eta, variants):
    """logLikelihood (HaploSNP_Sampler.py:431-442, Desman_Utils.py:23-33)."""
    p = np.einsum('ijk,lj,km->ilm', tau, gamma, eta)
    n = variants.sum(axis=2)
    return float((gammaln(n + 1) - gammaln(variants + 1).sum(axis=2)
                  + (variants * np.log(p)).sum(axis=2)).sum())


def log_dirichlet(x, alpha):
    """du.log_dirichlet_pdf (Desman_Utils.py:35-44)."""
    r = gammaln(np.sum(alpha))
    for i in range(len(alpha)):
        r += (alpha[i] - 1.0) * math.log(x[i])
        r -= gammaln(alpha[i])
    return float(r)


def log_posterior(tau, gamma, eta, variants, alpha=0.1, delta=0.1):
    """logPosterior (HaploSNP_Sampler.py:444-461)."""
    V, G, _ = tau.shape
    lp = log_likelihood(tau, gamma, eta, variants)
    for s in range(gamma.shape[0]):
        lp += log_dirichlet(gamma[s], np.full(G, alpha))
    for a in range(4):
        lp += log_dirichlet(eta[a], np.full(4, delta))
    return lp + V * G * math.log(1.0 / 4.0)


def gibbs_update(rs, tau, gamma, eta, variants, n_iter, sample_tau, alpha=0.1, delta=0.1,
                 epsilon=1e-6):
    """HaploSNP_Sampler.update (HaploSNP_Sampler.py:334-365).

    ``sample_tau(tau, gamma, eta, variants) -> nchange`` is the native boundary
    (sampletau.sample_tau), mutating ``tau`` in place.  Returns a dict with the
    final state, the star state and the per-iteration traces."""
    tau = tau.copy(); gamma = gamma.copy(); eta = eta.copy()
    ll = log_likelihood(tau, gamma, eta, variants)
    lp = log_posterior(tau, gamma, eta, variants, alpha, delta)
    star = dict(tau=tau.copy(), gamma=gamma.copy(), eta=eta.copy(), lp=lp, it=0)
    S, G = gamma.shape
    tr = dict(ll=np.zeros(n_iter), lp=np.zeros(n_iter), nchange=np.zeros(n_iter, dtype=np.int64),
              gamma=np.zeros((n_iter, S, G)), eta=np.zeros((n_iter, 4, 4)),
              tau_sum=np.zeros(tau.shape, dtype=np.int64), lp0=lp, ll0=ll)
    for it in range(n_iter):
        E, mu = sample_mu(rs, tau, gamma, eta, variants)
        gamma = sample_gamma(rs, mu, alpha, epsilon)
        nchange = sample_tau(tau, gamma, eta, variants)
        eta = sample_eta(rs, E, delta)
        ll = log_likelihood(tau, gamma, eta, variants)
        lp = log_posterior(tau, gamma, eta, variants, alpha, delta)
        tr['ll'][it] = ll; tr['lp'][it] = lp; tr['nchange'][it] = nchange
        if lp > star['lp']:
            star = dict(tau=tau.copy(), gamma=gamma.copy(), eta=eta.copy(), lp=lp, it=it)
        tr['gamma'][it] = gamma; tr['eta'][it] = eta; tr['tau_sum'] += tau
    return dict(tau=tau, gamma=gamma, eta=eta, star=star, trace=tr)


def calculate_snd(tau):
    """calculateSND (HaploSNP_Sampler.py:712-730): pairwise Hamming distance."""
    idx = np.argmax(tau, axis=2)
    return (idx[:, :, None] != idx[:, None, :]).sum(axis=0)


def remove_degenerate(tau, gamma):
    """removeDegenerate (HaploSNP_Sampler.py:771-800): merge h into g<h when
    their haplotypes are identical; gamma columns are summed; order kept."""
    G = tau.shape[1]
    snd = calculate_snd(tau)
    deleted = np.zeros(G, dtype=bool)
    merged = []
    for g in range(G):
        gm = []
        for h in range(g + 1, G):
            if not deleted[h] and snd[g, h] == 0:
                deleted[h] = True
                gm.append(h)
        merged.append(gm)
    keep = [g for g in range(G) if not deleted[g]]
    tau_new = np.ascontiguousarray(tau[:, keep, :])
    gamma_new = np.zeros((gamma.shape[0], len(keep)))
    for k, g in enumerate(keep):
        gamma_new[:, k] = gamma[:, g]
        for h in merged[g]:
            gamma_new[:, k] += gamma[:, h]
    return tau_new, gamma_new


# --------------------------------------------------------------------------
# Init_NMFT
# --------------------------------------------------------------------------
def nmft_freq(variants):
    """Init_NMFT.__init__ (Init_NMFT.py:49-60): F[v + a*V, s]."""
    x = variants.astype(np.float64) + 1.0
    f = x / x.sum(axis=2)[:, :, None]
    V, S, _ = variants.shape
    return np.ascontiguousarray(np.transpose(f, (2, 0, 1)).reshape(4 * V, S))


def nmft_random_initialize(rs, V, S, G, alpha=0.01):
    """random_initialize (Init_NMFT.py:66-78): gamma [G,S] then tau [4V,G] drawn
    v-major, g-minor.  dirichlet(a, size=n) equals n successive calls."""
    if G > 1:
        gamma = np.ascontiguousarray(rs.dirichlet(np.full(G, alpha), size=S).T)
    else:
        gamma = np.ones((G, S))
    return nmft_random_initialize_tau(rs, V, G, alpha), gamma


def nmft_random_initialize_tau(rs, V, G, alpha=0.01):
    """random_initialize_tau (Init_NMFT.py:80-86)."""
    d = rs.dirichlet(np.full(4, alpha), size=V * G).reshape(V, G, 4)
    return np.ascontiguousarray(np.transpose(d, (2, 0, 1)).reshape(4 * V, G))


def _elop_div(x, y):
    x = x.copy(); y = y.copy()
    x[x == 0] = EPS; y[y == 0] = EPS
    return x / y


def nmft_objective(F, tau, gamma):
    """div_objective (Init_NMFT.py:152-156)."""
    pa = np.maximum(tau @ gamma, EPS)
    return float((F * np.log(_elop_div(F, pa)) - F + pa).sum())


def nmft_update_tau(F, tau, gamma):
    """tau half of div_update == div_update_tau (Init_NMFT.py:170-181,192-205)."""
    N, G = tau.shape
    V = N // 4
    tau1 = np.tile(gamma.sum(1)[None, :], (N, 1))
    tau = tau * _elop_div(_elop_div(F, tau @ gamma) @ gamma.T, tau1)
    t3 = tau.reshape(4, V, G)
    tot = ((t3[0] + t3[1]) + t3[2]) + t3[3]            # sequential a = 0..3 from 0.0
    return (t3 / tot[None]).reshape(N, G)


def nmft_update(F, tau, gamma):
    """div_update (Init_NMFT.py:158-181)."""
    G, S = gamma.shape
    H1 = np.tile(tau.sum(0)[:, None], (1, S))
    if G > 1:
        gamma = gamma * _elop_div(tau.T @ _elop_div(F, tau @ gamma), H1)
        gamma = gamma / gamma.sum(axis=0)[None, :]
    else:
        gamma = np.ones((G, S))
    return nmft_update_tau(F, tau, gamma), gamma


def nmft_factorize(F, tau, gamma, max_iter=5000, min_change=1e-5):
    """factorize loop (Init_NMFT.py:98-115) after random_initialize."""
    tau = np.maximum(tau, EPS); gamma = np.maximum(gamma, EPS)
    divl, div, it = 0.0, nmft_objective(F, tau, gamma), 0
    trace = [div]
    while it < max_iter and math.fabs(divl - div) > min_change:
        tau, gamma = nmft_update(F, tau, gamma)
        tau = np.maximum(tau, EPS); gamma = np.maximum(gamma, EPS)
        divl, div = div, nmft_objective(F, tau, gamma)
        it += 1
        trace.append(div)
    return tau, gamma, it, np.array(trace)


def nmft_get_tau(tau, G):
    """get_tau (Init_NMFT.py:230-245): argmax with strict '>' from 0.0."""
    V = tau.shape[0] // 4
    t3 = tau.reshape(4, V, G)
    best = np.zeros((V, G)); arg = np.zeros((V, G), dtype=np.int64)
    for a in range(4):
        m = t3[a] > best
        best = np.where(m, t3[a], best)
        arg = np.where(m, a, arg)
    return cbind.idx_to_onehot(arg)


# --------------------------------------------------------------------------
# Variant_Filter likelihood-ratio filter (row f3)
# --------------------------------------------------------------------------
def fminbound(func, x1, x2, xatol=1e-5, maxfun=500):
    """Bounded Brent minimiser = scipy.optimize.minimize_scalar(method='bounded'), the third-party
    routine the reference calls at Variant_Filter.py:353 (SciPy, unpinned; restated from
    scipy/optimize/_optimize.py:_minimize_scalar_bounded and checked against the installed SciPy
    in tests/test_oracle_golden.py).  Returns the abscissa of the minimum."""
    sqrt_eps = math.sqrt(2.2e-16)
    golden_mean = 0.5 * (3.0 - math.sqrt(5.0))
    a, b = x1, x2
    fulc = a + golden_mean * (b - a)
    nfc, xf = fulc, fulc
    rat = e = 0.0
    x = xf
    fx = func(x)
    num = 1
    ffulc = fnfc = fx
    xm = 0.5 * (a + b)
    tol1 = sqrt_eps * abs(xf) + xatol / 3.0
    tol2 = 2.0 * tol1
    sgn = lambda z: (1.0 if z > 0 else -1.0 if z < 0 else 0.0) + (1.0 if z == 0 else 0.0)   # noqa: E731
    while abs(xf - xm) > (tol2 - 0.5 * (b - a)):
        golden = True
        if abs(e) > tol1:
            golden = False
            r = (xf - nfc) * (fx - ffulc)
            q = (xf - fulc) * (fx - fnfc)
            pp = (xf - fulc) * q - (xf - nfc) * r
            q = 2.0 * (q - r)
            if q > 0.0:
                pp = -pp
            q = abs(q)
            r = e
            e = rat
            if abs(pp) < abs(0.5 * q * r) and pp > q * (a - xf) and pp < q * (b - xf):
                rat = (pp + 0.0) / q
                x = xf + rat
                if (x - a) < tol2 or (b - x) < tol2:
                    rat = tol1 * sgn(xm - xf)
            else:
                golden = True
        if golden:
            e = a - xf if xf >= xm else b - xf
            rat = golden_mean * e
        x = xf + sgn(rat) * max(abs(rat), tol1)
        fu = func(x)
        num += 1
        if fu <= fx:
            if x >= xf:
                a = xf
            else:
                b = xf
            fulc, ffulc = nfc, fnfc
            nfc, fnfc = xf, fx
            xf, fx = x, fu
        else:
            if x < xf:
                a = x
            else:
                b = x
            if fu <= fnfc or nfc == xf:
                fulc, ffulc = nfc, fnfc
                nfc, fnfc = x, fu
            elif fu <= ffulc or fulc == xf or fulc == nfc:
                fulc, ffulc = x, fu
        xm = 0.5 * (a + b)
        tol1 = sqrt_eps * abs(xf) + xatol / 3.0
        tol2 = 2.0 * tol1
        if num >= maxfun:
            break
    return xf


def mix_nll(pm, eta, n, m, f):
    """mixNLL (Variant_Filter.py:38-41)."""
    return float(np.dot(f, -np.log(pm * eta[n, :] + (1 - pm) * eta[m, :])))


def lrt_step(ffreq, maxA, maxB, eta, upperP, optimise, p):
    """inner loop of get_filtered_VariantsLogRatio (Variant_Filter.py:348-356) -> (p, MLL, BLL)."""
    V = ffreq.shape[0]
    p = np.array(p, dtype=np.float64)
    MLL = np.zeros(V)
    BLL = -(np.log(eta[maxA, :]) * ffreq).sum(axis=1)
    for v in range(V):
        if optimise:
            p[v] = fminbound(lambda x: mix_nll(x, eta, maxA[v], maxB[v], ffreq[v]), 0.0, upperP)
        MLL[v] = mix_nll(p[v], eta, maxA[v], maxB[v], ffreq[v])
    return p, MLL, BLL
