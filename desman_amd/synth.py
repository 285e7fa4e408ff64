"""Synthetic V x S x 4 variant-count tensors for benchmarks and parity tests.

Generator specified in BASELINE.md sec. 3 item 4 / SURVEY.md sec. 8(d): the
reference ships no generator, so this is our own (numpy ``default_rng``).
"""
import numpy as np


def synth_counts(V, S, G, seed=1234):
    """Returns (counts int64 [V,S,4] C-order, tau_true uint8 [V,G], gamma_true [S,G]).

    tau_true ~ U{0..3} with forced variability (a position where every haplotype
    agrees gets haplotype 1 moved to another base), gamma_true ~ Dir(1_G) per
    sample, eta_true = 0.96 I + 0.01, depth ~ Poisson(c_s) with c_s ~ U(40, 500),
    counts ~ Multinomial(depth, sum_g gamma[s,g] eta[tau[v,g], :]).
    """
    rng = np.random.default_rng(seed)
    tau = rng.integers(0, 4, size=(V, G))
    if G > 1:
        same = (tau == tau[:, :1]).all(axis=1)
        tau[same, 1] = (tau[same, 1] + 1 + rng.integers(0, 3, size=int(same.sum()))) % 4
    gamma = rng.dirichlet(np.ones(G), size=S)
    eta = 0.96 * np.eye(4) + 0.01
    cs = rng.uniform(40, 500, size=S)
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
