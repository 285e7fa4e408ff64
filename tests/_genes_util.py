"""Shared set-up of the accessory-gene (f4) tests: the synth_genes() data set of a golden fixture, reshaped
the way GeneAssign.main feeds Eta_Sampler (tests/golden/make_golden.py: gen_gene_assign)."""
import ast
import os

import numpy as np

from desman_amd.synth import synth_genes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    kw = dict(ast.literal_eval(str(z['synth_kw'])))
    C, S, G = int(z['C']), int(z['S']), int(z['G'])
    d = synth_genes(C, S, G, seed=int(z['synth_seed']), **kw)
    gm = d['gamma'] / d['gamma'].sum(axis=1)[:, None]                 # GeneAssign.py:226-228
    delta = gm * d['total_mean'][:, None]                            # :230
    gene_off = np.concatenate([[0], np.cumsum(np.bincount(d['gene_of'], minlength=C))]).astype(np.int32)
    variants = [np.ascontiguousarray(d['counts'][gene_off[c]:gene_off[c + 1]]) for c in range(C)]
    return dict(z=z, d=d, C=C, S=S, G=G, gamma=np.ascontiguousarray(gm), delta=np.ascontiguousarray(delta),
                delta_gs=np.ascontiguousarray(delta.T), gene_off=gene_off, variants=variants,
                eps=np.ascontiguousarray(d['epsilon']), cov=np.ascontiguousarray(d['cov']),
                seed=int(z['seed']), iters=int(z['iters']), tau_iter=int(z['tau_iter']))


def split(cat, gene_off):
    return [np.ascontiguousarray(cat[gene_off[c]:gene_off[c + 1]]).astype(np.int64) for c in range(len(gene_off) - 1)]
