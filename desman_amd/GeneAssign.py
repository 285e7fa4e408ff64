"""`GeneAssign` on the GPU: copy number of every accessory gene / contig in every inferred haplotype.

Drop-in for desman/GeneAssign.py: same positional arguments, flags, defaults and output files
(<stub>etaD_df.csv, etaS_df.csv, etaM_df.csv, eta_df.csv and, with --assign_tau, <stub>_tau_star.csv /
_tau_mean.csv).  Stages: read the five tables -> KL non-negative fit of the coverages (kl_*_kernel) -> rounded
start -> two rounds of the copy-number Gibbs sampler (Eta_Sampler) -> optional per-variant haplotypes.

Extensions: --device, and --rng philox for the all-genes-at-once sampler (default: the reference's streams).
"""
import argparse
import logging
import os
import sys

import numpy as np
import pandas as pd
from numpy.random import RandomState

from . import _lib, sampletau
from .Eta_Sampler import Eta_Sampler
from .gene_shards import gather_blocks, partition_genes


def expand_sample_names(sample_names):
    return [name + suffix for name in sample_names for suffix in ("-A", "-C", "-G", "-T")]


class KLAssign:
    """cov [C,S] ~ eta [C,G] x delta^T under the generalised KL divergence, multiplicative updates on the
    device (GeneAssign.py:54-120).  The uniform start is drawn from the caller's RandomState."""

    def __init__(self, randomState, cov, delta, n_run=None, max_iter=None, min_change=None, device=0):
        self.name = "KLAssign"
        self.randomState = randomState
        self.cov = np.ascontiguousarray(cov, dtype=np.float64)
        self.delta = np.ascontiguousarray(delta, dtype=np.float64)          # [S,G]
        self.deltat = np.transpose(self.delta)
        self.C, self.G = self.cov.shape[0], self.delta.shape[1]
        self.n_run = 1 if n_run is None else n_run
        self.max_iter = 10000 if max_iter is None else max_iter
        self.min_change = 1.0e-4 if min_change is None else min_change
        self.device = device
        self.n_iter = 0
        self.div = None

    def random_initialize(self):
        self.eta = self.randomState.uniform(0, 1.0, (self.C, self.G))

    def factorize(self):
        for _ in range(self.n_run):
            self.random_initialize()
            self.eta, self.n_iter, self.div = _lib.kl_assign(self.cov, self.delta, self.eta, self.max_iter,
                                                             self.min_change, self.device)
            logging.info('KL fit: %d updates, divergence = %f' % (self.n_iter, self.div))

    def div_objective(self):
        _, _, div = _lib.kl_assign(self.cov, self.delta, self.eta, 0, self.min_change, self.device)
        return div


def compGenes(etaPred, etaG):
    """greedy one-to-one matching of predicted to known genomes by the fraction of genes they agree on;
    returns (mean accuracy of the matches, accuracy per known genome, the full agreement matrix)"""
    n_known, n_pred = etaG.shape[1], etaPred.shape[1]
    agree = (etaG[:, :, None] == etaPred[:, None, :]).sum(axis=0) / float(etaPred.shape[0])
    full = np.array(agree, copy=True)
    accuracies = np.zeros(n_known)
    total = 0.0
    for _ in range(n_known):
        r, col = np.unravel_index(np.argmax(agree), agree.shape)
        accuracies[r] = agree[r, col]
        total += agree[r, col]
        agree[r, :] = 0.0
        agree[:, col] = 0.0
    return (total / float(n_known), accuracies, full)


def build_parser():
    ap = argparse.ArgumentParser(prog="GeneAssign", description="accessory-gene copy numbers per haplotype (MI355X)")
    ap.add_argument("scg_cov_file", help="core-gene coverage per sample: columns mean, sd")
    ap.add_argument("gamma_star_file", help="haplotype abundances per sample (Gamma_star.csv of desman)")
    ap.add_argument("cov_file", help="mean coverage of every gene / contig per sample")
    ap.add_argument("epsilon_file", help="4x4 base transition matrix (Eta_star.csv of desman)")
    ap.add_argument('-s', '--random_seed', default=23724839, type=int, help="seed of both random streams")
    ap.add_argument('-e', '--eta_max', default=2, type=int, help="copy numbers 0..eta_max-1 are sampled")
    ap.add_argument('-i', '--iter_max', default=20, type=int, help="Gibbs iterations per round")
    ap.add_argument('-m', '--var_max', default=1e10, type=int, help="at most this many variant rows per gene")
    ap.add_argument('-o', '--output_stub', type=str, default="output", help="prefix of every output file")
    ap.add_argument('-g', '--genomes', help="known gene content, for an accuracy report")
    ap.add_argument('-v', '--variant_file', help="base counts at the variant positions of the genes")
    ap.add_argument('--assign_tau', dest='assign_tau', action='store_true', help="also write per-variant haplotypes")
    ap.add_argument('--device', type=int, default=0, help="GPU ordinal (extension)")
    ap.add_argument('--rng', choices=("mt19937", "philox"), default="mt19937",
                    help="mt19937: the reference's random streams; philox: all genes advance together (extension)")
    ap.set_defaults(assign_tau=False)
    return ap


def _write_haplotypes(path, values, index, positions):
    frame = pd.DataFrame(values.reshape(values.shape[0], -1), index=index)
    frame['Position'] = positions
    cols = frame.columns.tolist()
    frame[cols[-1:] + cols[:-1]].to_csv(path)


def main(argv=None):
    args = build_parser().parse_args(argv)
    stub = args.output_stub
    logging.basicConfig(filename=stub + "_log_file.txt", level=logging.INFO, filemode='w',
                        format='%(asctime)s:%(levelname)s:%(name)s:%(message)s')
    logging.info('seed of both random streams = %d' % (args.random_seed))
    prng = RandomState(args.random_seed)
    sampletau.initRNG()
    sampletau.setRNG(args.random_seed)

    read = lambda path: pd.read_csv(path, header=0, index_col=0)
    scg_cov, gamma_star, cov = read(args.scg_cov_file), read(args.gamma_star_file), read(args.cov_file)
    epsilon = read(args.epsilon_file).to_numpy()
    variants = read(args.variant_file) if args.variant_file is not None else None
    common = sorted(set(gamma_star.index.values) & set(scg_cov.index.values) & set(cov.columns.values))
    logging.info('%d samples are present in all three tables' % (len(common)))
    scg_cov, gamma_star, cov = scg_cov.reindex(common), gamma_star.reindex(common), cov[common]

    gamma = gamma_star.to_numpy()
    gamma = gamma / gamma.sum(axis=1)[:, np.newaxis]
    delta = gamma * scg_cov['mean'].to_numpy()[:, np.newaxis]               # expected coverage of one copy

    # ---- one process per GPU (python -m torch.distributed.run ...): the genes are sharded over the ranks
    rank, world, dist = 0, 1, None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            import torch
            dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
        rank, world = dist.get_rank(), dist.get_world_size()
        if args.rng != "philox":
            raise SystemExit("GeneAssign: sharding over ranks needs --rng philox (the reference's streams are serial)")
    device = args.device if world == 1 else int(os.environ.get("LOCAL_RANK", rank))

    logging.info('KL fit of the gene coverages')
    kl = KLAssign(prng, cov.to_numpy(), delta, device=device)           # cheap: every rank fits all genes
    kl.factorize()
    start = np.rint(kl.eta)

    gene_variants = variants[expand_sample_names(common)] if variants is not None else None
    names = cov.index.tolist()

    # ---- this rank's block of genes
    n_rows = np.zeros(len(names), dtype=np.int64)
    if gene_variants is not None:
        per_gene = pd.Series(np.ones(len(gene_variants), dtype=np.int64)).groupby(gene_variants.index.values).sum()
        n_rows = np.array([int(per_gene.get(g, 0)) for g in names], dtype=np.int64)
    bounds = partition_genes(n_rows, world)
    lo, hi = int(bounds[rank]), int(bounds[rank + 1])
    mine = names[lo:hi]
    my_variants, my_positions = gene_variants, variants
    if world > 1 and gene_variants is not None:
        keep = gene_variants.index.isin(set(mine))
        my_variants, my_positions = gene_variants[keep], variants[keep]
    sampler = Eta_Sampler(prng, my_variants, cov.iloc[lo:hi], gamma, delta, scg_cov['sd'].to_numpy(), epsilon, start[lo:hi],
                          max_iter=args.iter_max, max_eta=args.eta_max, max_var=args.var_max, device=device,
                          rng=args.rng, gene_base=lo)
    sampler.update()
    sampler.update()
    local = {"eta_star": sampler.eta_star, "eta_mean": np.mean(sampler.eta_store, axis=0)}
    if args.assign_tau is True:
        sampler.restoreFullVariants()
        sampler.calcTauStar(sampler.eta_star)
        tau_star, tau_mean, pos, owner = sampler.getTauStar(my_positions)
        local.update(tau_star=tau_star, tau_mean=tau_mean, pos=pos, owner=np.asarray(owner, dtype=object))
    tables = gather_blocks(local, dist)
    if rank != 0:
        return

    if args.assign_tau is True:
        _write_haplotypes(stub + "_tau_star.csv", tables["tau_star"], list(tables["owner"]), tables["pos"])
        logging.info("haplotypes of the gene variants written (MAP)")
        _write_haplotypes(stub + "_tau_mean.csv", tables["tau_mean"], list(tables["owner"]), tables["pos"])
        logging.info("haplotypes of the gene variants written (mean)")

    for suffix, table in (("etaD_df.csv", start), ("etaS_df.csv", tables["eta_star"]),
                          ("etaM_df.csv", tables["eta_mean"]), ("eta_df.csv", kl.eta)):
        pd.DataFrame(table, index=names).to_csv(stub + suffix)

    if args.genomes:
        known = read(args.genomes).loc[names].to_numpy()
        kl_total, _, _ = compGenes(start, known)
        gibbs_total, _, _ = compGenes(tables["eta_star"], known)
        logging.info('KL accurracy = %f' % (kl_total))
        logging.info('Gibbs sampler accurracy = %f' % (gibbs_total))


if __name__ == "__main__":
    main()
