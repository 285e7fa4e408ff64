"""`desman` command line on the device-resident classes.

Drop-in contract (reference: bin/desman:21-242): the same flags with the same defaults and quirks, and the
same output files.  The driver itself is organised differently: a flag table, then four
stages -- load, fit, report, assign-the-rest.
"""
import argparse
import logging
import os
import sys
import threading

import numpy as np
import pandas as pd
from numpy.random import RandomState

from . import sampletau
from .HaploSNP_Sampler import HaploSNP_Sampler
from .Init_NMFT import Init_NMFT
from .Output_Results import Output_Results
from .Variant_Filter import Variant_Filter

POSITION_SELECT_SEED = 238329          # fixed seed of the -r position draw (bin/desman:85-86)

# (short, long, kwargs) -- defaults and nargs/const quirks as in bin/desman:25-59
_FLAGS = [
    ('-g', '--genomes', dict(type=int, required=True, help="number of haplotypes to infer")),
    ('-f', '--filter_variants', dict(nargs='?', const=3.84, type=float,
                                     help="likelihood-ratio variant filter; optional chi2 threshold (3.84)")),
    ('-r', '--random_select', dict(nargs='?', const=1e3, type=int,
                                   help="fit on this many random positions (1000), then assign the others")),
    ('-e', '--eta_file', dict(type=str, help="CSV with the initial 4x4 error matrix")),
    ('-a', '--assign_file', dict(type=str, help="(dead upstream) extra positions to assign")),
    ('-o', '--output_dir', dict(type=str, default="output", help="directory for all result files")),
    ('-p', '--optimiseP', dict(default=True, type=bool, help="optimise the mixture proportion in the filter")),
    ('-i', '--no_iter', dict(nargs='?', const=250, type=int, help="Gibbs iterations per phase")),
    ('-m', '--min_coverage', dict(type=float, default=5.0, help="drop samples whose mean depth is not above this")),
    ('-q', '--max_qvalue', dict(default=1.0e-3, type=float, help="q-value cut of the variant filter")),
    ('-s', '--random_seed', dict(default=23724839, type=int, help="seed of both sampler RNG streams")),
    ('-v', '--min_variant_freq', dict(nargs='?', const=0.01, type=float, help="minimum variant frequency (0.01)")),
]


def build_parser():
    ap = argparse.ArgumentParser(prog="desman", description="strain haplotypes and abundances from base counts (MI355X)")
    ap.add_argument("variant_file", help="base-count table: Contig,Position,<sample>-A,-C,-G,-T,...")
    for short, long_, kw in _FLAGS:
        ap.add_argument(short, long_, **kw)
    ap.add_argument('--device', type=int, default=0, help="GPU ordinal (extension)")
    return ap


_TABLES_LOCK = threading.Lock()
_TABLES_CACHE = {}                      # (path, mtime, size) -> parsed frame, the most recent few


def _read_table(path):
    """the base-count table of `path`, parsed once per process while the file is unchanged: a G-sweep runs every chain through
    main() / main_replicates() in one process (desman_amd.chains), and parsing a 50 000 x 96 table (58 MB of text) costs more than
    the chain's GPU time.  The frame is only ever read (Variant_Filter copies it into arrays, Output_Results slices it)."""
    st = os.stat(path)
    key = (os.path.abspath(path), st.st_mtime_ns, st.st_size)
    with _TABLES_LOCK:
        table = _TABLES_CACHE.get(key)
        if table is None:
            table = pd.read_csv(path, header=0, index_col=0)
            while len(_TABLES_CACHE) >= 2:
                _TABLES_CACHE.pop(next(iter(_TABLES_CACHE)))
            _TABLES_CACHE[key] = table
    return table


def _load(opts, report):
    """CSV -> count tensor, sample filter, optional variant filter / eta file / position subsample."""
    logging.info('position-selection RNG seeded with %d' % POSITION_SELECT_SEED)
    table = _read_table(opts.variant_file)
    flt = Variant_Filter(table, randomState=RandomState(POSITION_SELECT_SEED), optimise=opts.optimiseP,
                         threshold=opts.filter_variants, min_coverage=opts.min_coverage, qvalue_cutoff=opts.max_qvalue)
    flt.device = opts.device
    if flt.S < 1 or flt.V < 1:
        logging.error('nothing to do: %d samples above the coverage cut, %d positions' % (flt.S, flt.V))
        sys.exit()
    logging.info('%d samples, %d positions, %d haplotypes requested' % (flt.S, flt.V, opts.genomes))
    if opts.filter_variants is not None:
        logging.info('variant filter: optimise=%s threshold=%s min_coverage=%s q<=%s min_freq=%s'
                     % (opts.optimiseP, opts.filter_variants, opts.min_coverage, opts.max_qvalue, opts.min_variant_freq))
        flt.get_filtered_VariantsLogRatio()                       # lrt_kernel on the GPU
        logging.info('variant filter kept %d positions' % flt.NS)
    if opts.eta_file is not None:
        logging.info('initial error matrix read from %s' % opts.eta_file)
        flt.eta = pd.read_csv(opts.eta_file, header=0, index_col=0).to_numpy()
    subsample = opts.random_select
    if subsample is not None:
        if subsample < flt.V:
            logging.info('fitting on %d random positions' % subsample)
            flt.select_Random(subsample)
        else:
            logging.info('only %d positions: the -r subsample of %d is ignored' % (flt.V, subsample))
            subsample = None
    return table, flt, subsample


def _fit(opts, flt):
    """NMF-tensor initialisation, burn-in, degenerate-haplotype merge, sampling (bin/desman:129-153)."""
    logging.info('sampler seed %d', opts.random_seed)
    rng = RandomState(opts.random_seed)
    sampletau.initRNG()
    sampletau.setRNG(opts.random_seed)
    nmft = Init_NMFT(flt.snps_filter, opts.genomes, rng, device=opts.device)
    logging.info('NMF-tensor initialisation')
    nmft.factorize()
    chain = HaploSNP_Sampler(flt.snps_filter, opts.genomes, rng, max_iter=opts.no_iter, device=opts.device,
                             ctx=nmft._ctx)                        # same resident count tensor
    chain.tau = np.copy(nmft.get_tau(), order='C')
    chain.updateTauIndices()
    chain.gamma = np.copy(nmft.get_gamma(), order='C')
    chain.eta = np.copy(flt.eta, order='C')
    logging.info('Gibbs burn-in')
    chain.update()
    chain.removeDegenerate()
    logging.info('Gibbs sampling')
    chain.update()
    return chain


# result tables of a fitted chain: (writer method of Output_Results, attribute or method of the sampler)
_TABLES = (("output_Filtered_Tau", "tau_star"), ("output_Tau_Mean", "tauMean"), ("output_Gamma", "gamma_star"),
           ("output_Gamma_Mean", "gammaMean"), ("output_Eta", "eta_star"), ("output_Eta_Mean", "etaMean"))


def _report(report, table, flt, chain, requested):
    report.set_Variants(table)
    report.set_Variant_Filter(flt)
    report.set_haplo_SNP(chain, requested)
    for writer, source in _TABLES:
        value = getattr(chain, source)
        getattr(report, writer)(value() if callable(value) else value)
    report.output_Selected_Variants()


def _assign_rest(opts, report, table, flt, chain):
    """-r: haplotypes of the positions left out of the fit, with tau-only sweeps driven by the fitted
    chain's gamma / eta traces (bin/desman:181-206)."""
    rest = flt.snps_filter_original[~np.asarray(flt.selected, dtype=bool), :]
    nmft = Init_NMFT(rest, chain.G, chain.randomState, device=opts.device)
    nmft.gamma = np.transpose(chain.gamma)
    logging.info('NMF-tensor initialisation of the %d remaining positions (gamma fixed)' % rest.shape[0])
    nmft.factorize_tau()
    other = HaploSNP_Sampler(rest, chain.G, chain.randomState, max_iter=opts.no_iter, device=opts.device, ctx=nmft._ctx)
    other.tau = nmft.get_tau()
    other.updateTauIndices()
    other.gamma_star = np.copy(chain.gammaMean(), order='C')
    other.eta_star = np.copy(chain.etaMean(), order='C')
    other.gamma_store = np.copy(chain.gamma_store, order='C')
    other.eta_store = np.copy(chain.eta_store, order='C')
    for phase in ('burn-in', 'sampling'):
        logging.info('tau-only %s' % phase)
        other.updateTau()
    report.outPredFit(other, opts.genomes)
    report.output_collated_Tau(other, table)


def _assign_rest_batch(runs, tell):
    """_assign_rest of several replicate chains with their NMF fits and tau-only sweeps batched (equal shapes assumed)"""
    from . import _lib
    others, nm = [], []
    for k, r in enumerate(runs):
        tell(k)
        opts, flt, chain = r["opts"], r["flt"], r["chain"]
        rest = flt.snps_filter_original[~np.asarray(flt.selected, dtype=bool), :]
        nmft = Init_NMFT(rest, chain.G, chain.randomState, device=opts.device)
        nmft.gamma = np.transpose(chain.gamma)
        logging.info('NMF-tensor initialisation of the %d remaining positions (gamma fixed)' % rest.shape[0])
        nm.append(nmft)
    Init_NMFT.factorize_tau_batch(nm)
    for k, (r, nmft) in enumerate(zip(runs, nm)):
        tell(k)
        nmft._log_trace(nmft.div_trace)
        opts, flt, chain = r["opts"], r["flt"], r["chain"]
        rest = flt.snps_filter_original[~np.asarray(flt.selected, dtype=bool), :]
        other = HaploSNP_Sampler(rest, chain.G, chain.randomState, max_iter=opts.no_iter, device=opts.device, ctx=nmft._ctx)
        other.mt_state = chain.mt_state                          # the tau-only sweeps continue the chain's GSL stream
        other.tau = nmft.get_tau()
        other.updateTauIndices()
        other.gamma_star = np.copy(chain.gammaMean(), order='C')
        other.eta_star = np.copy(chain.etaMean(), order='C')
        other.gamma_store = np.copy(chain.gamma_store, order='C')
        other.eta_store = np.copy(chain.eta_store, order='C')
        others.append(other)
    for phase in ('burn-in', 'sampling'):
        for k in range(len(runs)):
            tell(k)
            logging.info('tau-only %s' % phase)
        HaploSNP_Sampler.updateTau_batch(others, on_chain=lambda o: tell(others.index(o)))
    for k, (r, other) in enumerate(zip(runs, others)):
        tell(k)
        r["report"].outPredFit(other, r["opts"].genomes)
        r["report"].output_collated_Tau(other, r["table"])


def main(argv=None):
    opts = build_parser().parse_args(argv)
    if opts.assign_file is not None:
        # upstream this branch stops in ipdb.set_trace() after the whole run (bin/desman:213-214): there is nothing
        # to mirror, and the user should not pay for NMFT + 2 x no_iter Gibbs iterations to learn it
        sys.exit('desman: -a/--assign_file is not supported (dead in the reference: bin/desman:213-214)')
    if opts.eta_file is not None and not os.path.isfile(opts.eta_file):
        sys.exit("desman: can't open eta file '%s'" % opts.eta_file)
    if opts.genomes < 0:
        logging.error('the haplotype number must be positive, got %d' % opts.genomes)
        sys.exit(-1)
    report = Output_Results(opts.output_dir)
    table, flt, subsample = _load(opts, report)
    chain = _fit(opts, flt)
    _report(report, table, flt, chain, opts.genomes)
    if subsample is not None:
        _assign_rest(opts, report, table, flt, chain)
    sampletau.freeRNG()


def main_replicates(argv_list, on_chain=None):
    """The `desman` runs of several replicate chains (argv lists that differ in -s / -o only) with their Gibbs iterations
    batched: one set of kernel launches per iteration for all of them (HaploSNP_Sampler.update_batch).  Loading, the NMF
    start, the degenerate-haplotype merge and the result files are per chain, as in main(); replicates whose haplotype
    counts differ after the merge finish one by one.  ``on_chain(k)`` is called before chain k's own work (log routing).
    The mu/E pass of a batch is the aggregated sampler whatever the table size, so a chain's draws are those of a
    single run only where that run takes the aggregated pass too (same law either way)."""
    from . import _lib
    tell = on_chain if on_chain is not None else (lambda k: None)
    runs = []
    for k, argv in enumerate(argv_list):
        tell(k)
        opts = build_parser().parse_args(argv)
        if opts.assign_file is not None:
            sys.exit('desman: -a/--assign_file is not supported (dead in the reference: bin/desman:213-214)')
        report = Output_Results(opts.output_dir)
        table, flt, subsample = _load(opts, report)
        logging.info('sampler seed %d', opts.random_seed)
        rng = RandomState(opts.random_seed)
        nmft = Init_NMFT(flt.snps_filter, opts.genomes, rng, device=opts.device)
        runs.append(dict(opts=opts, report=report, table=table, flt=flt, subsample=subsample, rng=rng, nmft=nmft))
    # the NMF starts: together where the batched kernels apply (one shape, S <= 128, G <= 16), else one by one
    nm = [r["nmft"] for r in runs]
    done = False
    if len(nm) > 1 and len({(o.V, o.S, o.G) for o in nm}) == 1:
        try:
            Init_NMFT.factorize_batch(nm)
            done = True
        except _lib.DesmanHipError as e:
            logging.info('batched NMF start not available (%s): one by one' % e)
    for k, r in enumerate(runs):
        tell(k)
        logging.info('NMF-tensor initialisation')
        if done:
            r["nmft"]._log_trace(r["nmft"].div_trace)
        else:
            r["nmft"].factorize()
    for k, r in enumerate(runs):
        opts, flt, rng, nmft = r["opts"], r["flt"], r["rng"], r["nmft"]
        chain = HaploSNP_Sampler(flt.snps_filter, opts.genomes, rng, max_iter=opts.no_iter, device=opts.device, ctx=nmft._ctx)
        # (a chain's mu/E specification is its own in a batch too -- round 5 -- so a replicate that ends up alone, because its haplotype
        # count changed in removeDegenerate or its batched unit failed, draws what it would have drawn in the batch)
        chain.mt_state = _lib.mt_seed_state(opts.random_seed)       # what initRNG(); setRNG(seed) leave in the module's stream
        chain.tau = np.copy(nmft.get_tau(), order='C')
        chain.updateTauIndices()
        chain.gamma = np.copy(nmft.get_gamma(), order='C')
        chain.eta = np.copy(flt.eta, order='C')
        r["chain"] = chain
    chains = [r["chain"] for r in runs]

    def together(group):
        shapes = {(c.V, c.S, c.G) for c in group}
        if len(group) > 1 and len(shapes) == 1:
            HaploSNP_Sampler.update_batch(group, on_chain=lambda c: tell(chains.index(c)))
        else:
            for c in group:
                tell(chains.index(c))
                c.update()
    for k in range(len(runs)):
        tell(k)
        logging.info('Gibbs burn-in (batch of %d chains)' % len(runs))
    together(chains)
    for k, c in enumerate(chains):
        tell(k)
        c.removeDegenerate()
        logging.info('Gibbs sampling')
    by_g = {}
    for c in chains:
        by_g.setdefault(c.G, []).append(c)
    for group in by_g.values():
        together(group)
    for k, r in enumerate(runs):
        tell(k)
        _report(r["report"], r["table"], r["flt"], r["chain"], r["opts"].genomes)
    # -r: the positions left out of the fit -- batched where the replicates still agree in shape
    rest_runs = [r for r in runs if r["subsample"] is not None]
    idx = {id(r): k for k, r in enumerate(runs)}
    groups = {}
    for r in rest_runs:
        n_rest = int((~np.asarray(r["flt"].selected, dtype=bool)).sum())
        groups.setdefault((r["chain"].G, n_rest, r["chain"].S), []).append(r)
    for group in groups.values():
        batched = False
        if len(group) > 1:
            # whatever the batched attempt drew from the chains' numpy streams before it gave up is handed back
            states = [r["chain"].randomState.get_state() for r in group]
            try:
                _assign_rest_batch(group, lambda j, g=group: tell(idx[id(g[j])]))
                batched = True
            except _lib.DesmanHipError as e:
                for r, st in zip(group, states):
                    r["chain"].randomState.set_state(st)
                logging.info('batched -r path not available (%s): one by one' % e)
        if not batched:
            for r in group:
                tell(idx[id(r)])
                sampletau.initRNG()
                sampletau.setRNGState(r["chain"].mt_state)          # the tau-only sweeps continue the chain's GSL stream
                _assign_rest(r["opts"], r["report"], r["table"], r["flt"], r["chain"])
                sampletau.freeRNG()
    return chains


if __name__ == "__main__":
    main()
