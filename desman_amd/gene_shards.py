"""Sharding of the accessory-gene sampler over the GPUs of a node (one process per GPU).

Genes are independent given (gamma, epsilon, delta): each rank samples a contiguous block of genes with its own
`Eta_Sampler(rng="philox", gene_base=..., row_base=...)`; the counter-based draws are keyed by global gene / row
indices, so the result does not depend on the number of ranks.  No collective on the data path; one all_gather of
the per-gene tables at the end (RCCL on GPUs, gloo in the CPU tests).
"""
import numpy as np


def partition_genes(rows_per_gene, world, fixed_cost=4):
    """Contiguous blocks of genes with (almost) equal cost, cost(gene) = its variant rows + fixed_cost (the
    coverage terms and the draw itself).  Returns world + 1 gene indices; block r = [b[r], b[r + 1])."""
    cost = np.asarray(rows_per_gene, dtype=np.float64) + float(fixed_cost)
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    bounds = [0]
    for r in range(1, world):
        target = cum[-1] * r / world
        b = int(np.searchsorted(cum, target, side="left"))
        bounds.append(min(max(b, bounds[-1]), len(cost)))
    bounds.append(len(cost))
    return np.asarray(bounds, dtype=np.int64)


def gather_blocks(local, dist=None):
    """all ranks' per-gene arrays (dict name -> array with genes / rows on axis 0), concatenated in rank order
    on every rank.  dist = torch.distributed (initialised) or None for a single process."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return {k: np.asarray(v) for k, v in local.items()}
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, {k: np.asarray(v) for k, v in local.items()})
    return {k: np.concatenate([p[k] for p in parts], axis=0) for k in local}
