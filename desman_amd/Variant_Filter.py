"""Host mirror of desman/Variant_Filter.py: the loader semantics the Gibbs path needs
(reshape the CSV frame into snps[V,S,4], drop low-coverage samples, default eta,
selection bookkeeping, `select_Random`) and -- SURVEY sec. 8 row f3 -- the
likelihood-ratio variant filter (`-f`, Variant_Filter.py:320-390), whose
per-variant bounded minimisation runs in the `lrt_kernel` HIP kernel
(dsm_lrt_step) instead of a Python loop over scipy.optimize.minimize_scalar.
Not mirrored: the COG outlier filters (`-c` of the stand-alone tool).
"""
import argparse
import logging
import os
import sys

import numpy as np
import pandas as p
from scipy.special import erf

from . import _lib


def benjamini_Hochberg(pvalues):
    """BH step-up q-values exactly as Variant_Filter.py:43-58 computes them: p-values ranked by
    (value, index) descending, q = p * n / rank, made monotone from the top."""
    pvalues = np.asarray(pvalues, dtype=np.float64)
    n = pvalues.shape[0]
    order = sorted(range(n), key=lambda i: (pvalues[i], i), reverse=True)
    q = np.array([(float(n) / (n - k)) * pvalues[i] for k, i in enumerate(order)])
    if n > 1:
        q = np.minimum.accumulate(q)
    out = np.zeros(n)
    out[order] = q
    return out


def addPositions(dataFrame, position):
    dataFrame['Position'] = position
    cols = dataFrame.columns.tolist()
    cols[0] = "value"
    dataFrame.columns = cols
    cols = cols[-1:] + cols[:-1]
    return dataFrame[cols]


class Variant_Filter:

    def __init__(self, variants, randomState, optimise=True, threshold=3.84, min_coverage=5.0,
                 qvalue_cutoff=0.1, max_iter=100, min_p=0.01, mCogFilter=2.0, cogSampleFrac=0.95,
                 Nthreshold=10):
        m = variants.to_numpy()
        self.genes = list(variants.index)
        self.position = m[:, 0]                                    # first data column = Position (:75-77)
        m = np.delete(m, 0, 1)
        snps = np.reshape(m, (m.shape[0], m.shape[1] // 4, 4))
        self.randomState = randomState
        depth = snps.sum(axis=2)
        self.sample_filter = np.mean(depth, axis=0) > min_coverage   # (:87)
        self.sample_indices = np.where(self.sample_filter)[0].tolist()
        self.snps_filter = snps[:, self.sample_filter, :]
        self.V, self.S = self.snps_filter.shape[0], self.snps_filter.shape[1]
        self.freq = self.snps_filter.sum(axis=1)
        self.ffreq = self.freq.astype(np.float64)
        self.threshold = threshold
        self.qvalue_cutoff = qvalue_cutoff
        self.optimise = optimise
        self.filtered = np.zeros(self.V, dtype=bool)
        self.max_iter = max_iter
        self.Nthreshold = Nthreshold
        self.mCogFilter = mCogFilter
        self.cogSampleFrac = cogSampleFrac
        self.device = 0
        self.eta = 0.96 * np.identity(4) + 0.01 * np.ones((4, 4))    # (:108)
        self.upperP = 1.0 - min_p
        self.NS = self.V
        self.selected = np.ones(self.V, dtype=bool)
        self.selected_indices = np.where(self.selected)[0].tolist()
        self.randomSelect = False

    def get_filtered_VariantsLogRatio(self):
        """likelihood-ratio filter (Variant_Filter.py:320-390): one-base model vs a two-base mixture
        at every position, error matrix re-estimated from the rejected positions until the
        selection stops changing, then chi2(1) p-values, BH q-values, cut at qvalue_cutoff."""
        it = 0
        self.maxA = np.argmax(self.freq, axis=1)
        ftemp = np.copy(self.freq)
        ftemp[np.arange(self.V), self.maxA] = -1
        self.maxB = np.argmax(ftemp, axis=1)
        N = self.freq.sum(axis=1).astype(np.float64)
        n = self.freq.max(axis=1).astype(np.float64)
        m = ftemp.max(axis=1).astype(np.float64)
        self.filtered = N < self.Nthreshold
        keep = ~self.filtered
        self.minV = np.zeros(self.V)
        self.minV[keep] = m[keep] / N[keep]
        pv = np.zeros(self.V)
        pv[keep] = n[keep] / N[keep]
        pv[pv > self.upperP] = self.upperP
        lastSelect, Select = 0, self.V
        ratioNLL = np.zeros(self.V)
        while it < self.max_iter and lastSelect != Select:
            pv, MLL, BLL = _lib.lrt_step(self.ffreq, self.maxA, self.maxB, self.eta, self.upperP, self.optimise, pv,
                                         device=self.device)                       # (:348-356) on the GPU
            ratioNLL = 2.0 * (BLL - MLL)
            self.filtered = np.logical_or(N < self.Nthreshold, ratioNLL < self.threshold)
            eta = 96 * np.identity(4) + np.ones((4, 4))
            np.add.at(eta, self.maxA[self.filtered], self.freq[self.filtered].astype(np.float64))
            self.eta = eta / eta.sum(axis=1)[:, np.newaxis]
            lastSelect = Select
            Select = self.V - self.filtered.sum()
            logging.info("Variant filter iter: " + str(it) + " " + str(Select) + " " + str(self.eta))
            it += 1
        self.pvalue = 1.0 - erf(np.sqrt(np.maximum(ratioNLL, 0.0) / 2.0))           # 1 - chi2.cdf(x, 1)
        self.pvalue[ratioNLL < 0] = 1.0
        self.qvalue = benjamini_Hochberg(self.pvalue)
        self.ratioNLL = ratioNLL
        self.filtered = np.logical_or(N < self.Nthreshold, self.qvalue > self.qvalue_cutoff)
        self.snps_filter = self.snps_filter[self.filtered != True, :, :]             # noqa: E712
        self.selected_indices = np.where(self.filtered != True)[0].tolist()          # noqa: E712
        self.selected = self.filtered != True                                        # noqa: E712
        self.NS = self.snps_filter.shape[0]
        return self.snps_filter

    def calc_Error_Matrix(self):
        """Laplace-smoothed base transition counts over the rejected positions (:412-429)."""
        tm = np.ones((4, 4))
        sbv = self.freq[self.filtered]
        np.add.at(tm, np.argmax(sbv, axis=1), sbv.astype(np.float64))
        self.tran_Matrix = tm / tm.sum(axis=1)[:, np.newaxis]
        return self.tran_Matrix

    def selected_variants_todf(self, variants):
        """the selected positions as a .freq-style frame (:229-259)."""
        snps = np.reshape(self.snps_filter, (self.NS, self.S * 4))
        position = self.position[self.selected]
        names = [g for g, keep in zip(self.genes, self.selected) if keep]
        cols = variants.columns.values.tolist()
        sample_cols = [cols[i * 4 + 1 + a] for i in self.sample_indices for a in range(4)]
        df = p.DataFrame(snps, index=names, columns=sample_cols)
        df['Position'] = p.Series(position, index=df.index)
        c = df.columns.tolist()
        return df[c[-1:] + c[:-1]]

    def select_Random(self, random_select):
        """sorted choice without replacement from the filter's RandomState (:392-410)."""
        if random_select < self.NS:
            self.randomSelect = True
            select = np.sort(self.randomState.choice(self.NS, random_select, replace=False))
            self.snps_filter_original = np.copy(self.snps_filter)
            self.snps_filter = self.snps_filter[select, :, :]
            self.NS = random_select
            self.selected_indices_original = np.copy(self.selected_indices)
            self.selected_indices = [self.selected_indices[i] for i in select]
            self.selected_original = np.copy(self.selected)
            self.selected = np.zeros(self.V, dtype=bool)
            self.selected[self.selected_indices] = True
        return self.snps_filter


def main(argv=None):
    """stand-alone variant filter (Variant_Filter.py:436-558): writes <stub>sel_var.csv, v_df, p_df, q_df,
    r_df and tran_df.csv."""
    parser = argparse.ArgumentParser()
    parser.add_argument("variant_file", help="input SNP frequencies")
    parser.add_argument('-f', '--filter_variants', nargs='?', const=3.84, type=float)
    parser.add_argument('-q', '--max_qvalue', nargs='?', const=1.0e-3, type=float)
    parser.add_argument('-v', '--min_variant_freq', nargs='?', const=0.01, type=float)
    parser.add_argument('-m', '--min_coverage', type=float, default=5.0)
    parser.add_argument('-o', '--output_stub', type=str, default="output")
    parser.add_argument('-p', '--optimiseP', action='store_true')
    parser.add_argument('-s', '--random_seed', default=23724839, type=int)
    parser.add_argument('--device', type=int, default=0)
    args = parser.parse_args(argv)
    stub = args.output_stub
    logging.basicConfig(filename=stub + 'log.txt', level=logging.INFO, filemode='w',
                        format='%(asctime)s:%(levelname)s:%(name)s:%(message)s')
    logging.info("Results created at {0}".format(os.path.abspath(stub + 'log.txt')))
    prng = np.random.RandomState(args.random_seed)
    max_qvalue = 1.0e-3 if args.max_qvalue is None else args.max_qvalue
    filter_variants = 25.0 if args.filter_variants is None else args.filter_variants
    min_variant_freq = 0.01 if args.min_variant_freq is None else args.min_variant_freq
    variants = p.read_csv(args.variant_file, header=0, index_col=0)
    vf = Variant_Filter(variants, randomState=prng, optimise=args.optimiseP, threshold=filter_variants,
                        min_coverage=args.min_coverage, qvalue_cutoff=max_qvalue, min_p=min_variant_freq)
    vf.device = args.device
    logging.info('Begun filtering variants with parameters: optimise probability = %s, lr threshold = %s, min. coverage = %s, '
                 'q-value threshold = %s, min. variant frequency = %s'
                 % (args.optimiseP, filter_variants, args.min_coverage, max_qvalue, min_variant_freq))
    vf.get_filtered_VariantsLogRatio()
    logging.info("Completed variant filtering")
    tm = vf.calc_Error_Matrix()
    vf.selected_variants_todf(variants).to_csv(stub + "sel_var.csv")
    for name, arr in (("v_df", vf.minV), ("p_df", vf.pvalue), ("q_df", vf.qvalue), ("r_df", vf.ratioNLL)):
        addPositions(p.DataFrame(arr, index=vf.genes), vf.position).to_csv(stub + name + ".csv")
    p.DataFrame(tm).to_csv(stub + "tran_df.csv")


if __name__ == "__main__":
    main(sys.argv[1:])
