"""Synthetic V x S x 4 variant-count tensors for benchmarks and parity tests.

Generator specified in BASELINE.md sec. 3 item 4 / SURVEY.md sec. 8(d): the
reference ships no generator, so this is our own (numpy ``default_rng``).
"""
import numpy as np


def synth_counts(V, S, G, seed=1234, eta=None, depth_scale=1.0):
    """Returns (counts int64 [V,S,4] C-order, tau_true uint8 [V,G], gamma_true [S,G]).

    tau_true ~ U{0..3} with forced variability (a position where every haplotype
    agrees gets haplotype 1 moved to another base), gamma_true ~ Dir(1_G) per
    sample, eta_true = 0.96 I + 0.01, depth ~ Poisson(c_s) with c_s ~ U(40, 500),
    counts ~ Multinomial(depth, sum_g gamma[s,g] eta[tau[v,g], :]).
    ``eta`` replaces the error matrix (rows = true base); ``depth_scale`` multiplies the mean depths
    (same random stream otherwise).
    """
    rng = np.random.default_rng(seed)
    tau = rng.integers(0, 4, size=(V, G))
    if G > 1:
        same = (tau == tau[:, :1]).all(axis=1)
        tau[same, 1] = (tau[same, 1] + 1 + rng.integers(0, 3, size=int(same.sum()))) % 4
    gamma = rng.dirichlet(np.ones(G), size=S)
    eta = 0.96 * np.eye(4) + 0.01 if eta is None else np.asarray(eta, dtype=np.float64)
    cs = rng.uniform(40, 500, size=S) * depth_scale
    depth = rng.poisson(np.broadcast_to(cs, (V, S)))
    p = np.einsum('sg,vgb->vsb', gamma, eta[tau])
    p = p / p.sum(axis=2, keepdims=True)
    counts = rng.multinomial(depth, p).astype(np.int64)
    return np.ascontiguousarray(counts), tau.astype(np.uint8), gamma


def random_state(V, S, G, seed=0):
    """A random (tau one-hot int64 [V,G,4], gamma [S,G], eta [4,4]) sampler state."""
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, 4, size=(V, G))
    tau = np.zeros((V, G, 4), dtype=np.int64)
    np.put_along_axis(tau, idx[..., None], 1, axis=2)
    gamma = rng.dirichlet(np.ones(G), size=S)
    eta = rng.dirichlet(np.array([1.0, 1.0, 1.0, 1.0]), size=4) * 0.08 + 0.92 * np.eye(4)
    return tau, np.ascontiguousarray(gamma), np.ascontiguousarray(eta)


def synth_genes(C, S, G, seed=77, vmax=12, empty_every=5, mean_lo=30.0, mean_hi=120.0):
    """Accessory-gene data set for Eta_Sampler / GeneAssign (desman/GeneAssign.py inputs).

    Returns a dict: ``genes`` (names), ``eta_true`` int [C,G] (gene g-copy number in each strain, 0/1, every
    gene in at least one strain), ``gamma`` [S,G], ``total_mean`` [S] (core-gene coverage), ``delta`` [S,G] =
    gamma * total_mean, ``cov`` float [C,S] (gene coverages ~ Poisson(eta delta^T) + small noise, real valued),
    ``epsilon`` [4,4], ``gene_of`` int [Vtot] (gene index of every variant row), ``pos`` int [Vtot],
    ``counts`` int64 [Vtot,S,4].  The core-gene coverage is U(mean_lo, mean_hi) per sample (a low range gives a
    flat posterior, i.e. a chain that actually moves).  Every ``empty_every``-th gene has no variant rows; gene 1 has exactly one.
    """
    rng = np.random.default_rng(seed)
    eta_true = (rng.random((C, G)) < 0.6).astype(np.int64)
    for c in range(C):
        if eta_true[c].sum() == 0:
            eta_true[c, rng.integers(0, G)] = 1
    gamma = rng.dirichlet(np.ones(G) * 2.0, size=S)
    total_mean = rng.uniform(mean_lo, mean_hi, size=S)
    delta = gamma * total_mean[:, None]
    lam = eta_true @ delta.T
    cov = rng.poisson(lam).astype(np.float64) + rng.uniform(0.0, 0.5, size=lam.shape)
    epsilon = 0.96 * np.eye(4) + 0.01
    n_var = rng.integers(2, vmax + 1, size=C)
    n_var[empty_every - 1::empty_every] = 0
    if C > 1:
        n_var[1] = 1
    gene_of = np.repeat(np.arange(C), n_var)
    Vtot = int(n_var.sum())
    pos = np.concatenate([np.sort(rng.choice(2000, size=n, replace=False)) for n in n_var]).astype(np.int64) \
        if Vtot else np.zeros(0, dtype=np.int64)
    tau = rng.integers(0, 4, size=(Vtot, G))
    mask = eta_true[gene_of].astype(np.float64)                       # [Vtot,G]
    gm = gamma[None, :, :] * mask[:, None, :]                           # [Vtot,S,G]
    gm = gm / gm.sum(axis=2, keepdims=True)
    p = np.einsum('vsg,vgb->vsb', gm, epsilon[tau])
    p = p / p.sum(axis=2, keepdims=True)
    depth = rng.poisson(np.maximum(lam[gene_of], 1.0))
    counts = rng.multinomial(depth, p).astype(np.int64)
    return dict(genes=["gene%03d" % c for c in range(C)], eta_true=eta_true, gamma=np.ascontiguousarray(gamma),
                total_mean=total_mean, delta=np.ascontiguousarray(delta), cov=cov, epsilon=epsilon,
                gene_of=gene_of, pos=pos, counts=np.ascontiguousarray(counts))


def write_gene_inputs(d, out_dir, total_sd=None):
    """Writes a synth_genes() data set as the five CSV inputs of GeneAssign (scg coverage, gamma, gene coverage,
    epsilon, gene variants); returns their paths in the CLI's positional order + the variant file."""
    import os
    import pandas as pd
    S = d['gamma'].shape[0]
    samples = ["sample%02d" % s for s in range(S)]
    sd = np.full(S, 1.0) if total_sd is None else total_sd
    paths = [os.path.join(out_dir, n) for n in ("scg_cov.csv", "gamma_star.csv", "gene_cov.csv", "epsilon.csv",
                                                "gene_variants.csv")]
    pd.DataFrame({'mean': d['total_mean'], 'sd': sd}, index=samples).to_csv(paths[0])
    pd.DataFrame(d['gamma'], index=samples).to_csv(paths[1])
    pd.DataFrame(d['cov'], index=d['genes'], columns=samples).to_csv(paths[2])
    pd.DataFrame(d['epsilon']).to_csv(paths[3])
    cols = [s + "-" + b for s in samples for b in "ACGT"]
    V = d['counts'].shape[0]
    frame = pd.DataFrame(d['counts'].reshape(V, S * 4), index=[d['genes'][c] for c in d['gene_of']], columns=cols)
    frame.insert(0, 'Position', d['pos'])
    frame.to_csv(paths[4])
    return paths
