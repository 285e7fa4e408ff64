"""`desman` command line: the flags, defaults, quirks and output files of the
reference's bin/desman (:21-242), driving the device-resident classes."""
import argparse
import logging
import sys

import numpy as np
import pandas as p
from numpy.random import RandomState

from . import HaploSNP_Sampler as hsnp
from . import Init_NMFT as inmft
from . import Output_Results as outr
from . import Variant_Filter as vf
from . import sampletau


def build_parser():
    parser = argparse.ArgumentParser(prog="desman")
    parser.add_argument("variant_file", help="input SNP frequencies")
    parser.add_argument('-g', '--genomes', type=int, required=True, help="specify the haplotype number")
    parser.add_argument('-f', '--filter_variants', nargs='?', const=3.84, type=float,
                        help='filters variants by negative binomial loge likelihood defaults to 3.84')
    parser.add_argument('-r', '--random_select', nargs='?', const=1e3, type=int,
                        help="selects subset of variants passing filter to build model and assigns others")
    parser.add_argument('-e', '--eta_file', type=open, help="reads initial eta matrix from file")
    parser.add_argument('-a', '--assign_file', type=open,
                        help="calculates haplotype profiles for these SNPs using fitted gamma, eta values")
    parser.add_argument('-o', '--output_dir', type=str, default="output",
                        help="string specifying output directory and file stubs")
    parser.add_argument('-p', '--optimiseP', default=True, type=bool,
                        help="optimise proportions in likelihood ratio test")
    parser.add_argument('-i', '--no_iter', nargs='?', const=250, type=int,
                        help='Number of iterations of Gibbs sampler')
    parser.add_argument('-m', '--min_coverage', type=float, default=5.0,
                        help='minimum coverage for sample to be included')
    parser.add_argument('-q', '--max_qvalue', default=1.0e-3, type=float,
                        help="specifies q value cut-off for variant detection defaults 1.0e-3")
    parser.add_argument('-s', '--random_seed', default=23724839, type=int,
                        help="specifies seed for numpy random number generator defaults to 23724839 applied after random filtering")
    parser.add_argument('-v', '--min_variant_freq', nargs='?', const=0.01, type=float,
                        help="specifies minimum variant frequency defaults 0.01")
    # extensions (not in the reference)
    parser.add_argument('--device', type=int, default=0, help="GPU ordinal (extension)")
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    genomes = args.genomes
    if genomes < 0:
        logging.error('Only positive haplotype number valid not  %d. Exiting!' % genomes)
        sys.exit(-1)
    no_iter = args.no_iter
    random_select = args.random_select

    output_Results = outr.Output_Results(args.output_dir)
    logging.info('Set fixed seed for random position selection = 238329')
    prng = RandomState(238329)
    variants = p.read_csv(args.variant_file, header=0, index_col=0)
    variant_Filter = vf.Variant_Filter(variants, randomState=prng, optimise=args.optimiseP,
                                       threshold=args.filter_variants, min_coverage=args.min_coverage,
                                       qvalue_cutoff=args.max_qvalue)
    if variant_Filter.S < 1 or variant_Filter.V < 1:
        logging.error('Not enough samples with minimum coverage %d or variant positions %d. Exiting!'
                      % (variant_Filter.S, variant_Filter.V))
        sys.exit()
    logging.info('Running Desman with %d samples and %d variant positions finding %d genomes.'
                 % (variant_Filter.S, variant_Filter.V, genomes))
    variant_Filter.device = args.device
    if args.filter_variants is not None:
        logging.info('Begun filtering variants with parameters: optimise probability = %s, lr threshold = %s, min. coverage = %s, q-value threshold = %s, min. variant frequency = %s'
                     % (args.optimiseP, args.filter_variants, args.min_coverage, args.max_qvalue, args.min_variant_freq))
        variant_Filter.get_filtered_VariantsLogRatio()          # row f3: lrt_kernel on the GPU
        logging.info("Completed variant filtering")
    if args.eta_file is not None:
        logging.info('Set eta error transition matrix from = %s' % args.eta_file)
        variant_Filter.eta = p.read_csv(args.eta_file, header=0, index_col=0).to_numpy()
    if random_select is not None:
        if random_select < variant_Filter.V:
            logging.info('Selected %d random variant positions to infer haplotypes from' % random_select)
            variant_Filter.select_Random(random_select)
        else:
            logging.info('Not enough variable positions for random selection %d >= %d using all'
                         % (random_select, variant_Filter.V))
            random_select = None

    logging.info('Set second adjustable random seed = %d', args.random_seed)
    prng = RandomState(args.random_seed)
    sampletau.initRNG()
    sampletau.setRNG(args.random_seed)

    init_NMFT = inmft.Init_NMFT(variant_Filter.snps_filter, genomes, prng, device=args.device)
    logging.info('Perform NTF initialisation')
    init_NMFT.factorize()

    haplo_SNP = hsnp.HaploSNP_Sampler(variant_Filter.snps_filter, genomes, prng, max_iter=no_iter,
                                      device=args.device, ctx=init_NMFT._ctx)      # same resident count tensor
    haplo_SNP.tau = np.copy(init_NMFT.get_tau(), order='C')
    haplo_SNP.updateTauIndices()
    haplo_SNP.gamma = np.copy(init_NMFT.get_gamma(), order='C')
    haplo_SNP.eta = np.copy(variant_Filter.eta, order='C')

    logging.info('Start Gibbs sampler burn-in phase')
    haplo_SNP.update()
    haplo_SNP.removeDegenerate()
    logging.info('Start Gibbs sampler sampling phase')
    haplo_SNP.update()

    output_Results.set_Variants(variants)
    output_Results.set_Variant_Filter(variant_Filter)
    output_Results.set_haplo_SNP(haplo_SNP, genomes)
    output_Results.output_Filtered_Tau(haplo_SNP.tau_star)
    output_Results.output_Tau_Mean(haplo_SNP.tauMean())
    output_Results.output_Gamma(haplo_SNP.gamma_star)
    output_Results.output_Gamma_Mean(haplo_SNP.gammaMean())
    output_Results.output_Eta(haplo_SNP.eta_star)
    output_Results.output_Eta_Mean(haplo_SNP.etaMean())
    output_Results.output_Selected_Variants()

    if random_select is not None:
        snps_notselected = variant_Filter.snps_filter_original[variant_Filter.selected != True, :]   # noqa: E712
        init_NMFT_NS = inmft.Init_NMFT(snps_notselected, haplo_SNP.G, haplo_SNP.randomState, device=args.device)
        init_NMFT_NS.gamma = np.transpose(haplo_SNP.gamma)
        logging.info('Perform NTF initialisation on not selected SNPs fixed gamma')
        init_NMFT_NS.factorize_tau()
        haplo_SNP_NS = hsnp.HaploSNP_Sampler(snps_notselected, haplo_SNP.G, haplo_SNP.randomState,
                                             max_iter=no_iter, device=args.device, ctx=init_NMFT_NS._ctx)
        haplo_SNP_NS.tau = init_NMFT_NS.get_tau()
        haplo_SNP_NS.updateTauIndices()
        haplo_SNP_NS.gamma_star = np.copy(haplo_SNP.gammaMean(), order='C')
        haplo_SNP_NS.eta_star = np.copy(haplo_SNP.etaMean(), order='C')
        haplo_SNP_NS.gamma_store = np.copy(haplo_SNP.gamma_store, order='C')
        haplo_SNP_NS.eta_store = np.copy(haplo_SNP.eta_store, order='C')
        logging.info('Start Gibbs sampler burn-in phase')
        haplo_SNP_NS.updateTau()
        logging.info('Start Gibbs sampler sampling phase')
        haplo_SNP_NS.updateTau()
        output_Results.outPredFit(haplo_SNP_NS, genomes)
        output_Results.output_collated_Tau(haplo_SNP_NS, variants)

    if args.assign_file is not None:
        # the reference stops in ipdb.set_trace() here (bin/desman:213-214): the path is dead upstream
        logging.error('-a/--assign_file is not supported (dead in the reference: ipdb breakpoint)')
        sys.exit('desman: -a/--assign_file is not supported')

    sampletau.freeRNG()


if __name__ == "__main__":
    main()
