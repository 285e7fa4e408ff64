"""Writers for DESMAN's output files (desman/Output_Results.py) -- the on-disk
contract of the drop-in CLI: same file names, same CSV layouts."""
import logging
import os
import sys

import numpy as np
import pandas as p


def rchop(s, ending):
    return s[:-len(ending)] if s.endswith(ending) else s


class Output_Results:

    def __init__(self, outputDir):
        self.outputDir = outputDir
        os.makedirs(outputDir, exist_ok=True)
        self.log_file_name = self.outputDir + "/log_file.txt"
        logging.basicConfig(filename=self.log_file_name, level=logging.INFO, filemode='w',
                            format='%(asctime)s:%(levelname)s:%(name)s:%(message)s')
        logging.info("Results created in {0}".format(os.path.abspath(self.outputDir)))
        print("Up and running. Check {0} for progress".format(os.path.abspath(self.log_file_name)), file=sys.stderr)

    def set_Variants(self, variants):
        self.variants = variants
        self.contig_names = variants.index.tolist()
        self.position = variants['Position']

    def set_Variant_Filter(self, variantFilter):
        self.variantFilter = variantFilter
        sel = variantFilter.selected_indices
        self.filtered_contig_names = [self.contig_names[i] for i in sel]
        self.filtered_position = [self.position.iloc[i] for i in sel]

    def _fit(self, name, haplo_SNP, genomes):
        with open(self.outputDir + "/" + name, "w") as f:
            f.write("Fit,%d,%d,%f,%f\n" % (genomes, haplo_SNP.G, haplo_SNP.lp_star, haplo_SNP.meanDeviance()))

    def set_haplo_SNP(self, haplo_SNP, genomes):
        self.haplo_SNP = haplo_SNP
        self._fit("fit.txt", haplo_SNP, genomes)
        logging.info("Wrote fit stats")

    def outPredFit(self, haplo_SNP, genomes):
        self._fit("fitP.txt", haplo_SNP, genomes)
        logging.info("Wrote pred fit stats")

    @staticmethod
    def _position_first(values, index, position):
        df = p.DataFrame(values, index=index)
        df['Position'] = position
        cols = df.columns.tolist()
        return df[cols[-1:] + cols[:-1]]

    def _tau_csv(self, name, tau):
        flat = np.reshape(tau, (self.haplo_SNP.V, self.haplo_SNP.G * 4))
        self._position_first(flat, self.filtered_contig_names, self.filtered_position).to_csv(self.outputDir + "/" + name)

    def output_Filtered_Tau(self, tau):
        self._tau_csv("Filtered_Tau_star.csv", tau)
        logging.info("Wrote filtered tau star haplotype predictions")

    def output_Tau_Mean(self, tauProb):
        self._tau_csv("Tau_Mean.csv", tauProb)
        logging.info("Wrote probabilistic tau haplotype predictions")

    def output_collated_Tau(self, haplo_SNP_NS, full_variants):
        VS = haplo_SNP_NS.V + self.haplo_SNP.V
        G = self.haplo_SNP.G
        sel = np.asarray(self.variantFilter.selected[:VS], dtype=bool)
        star = np.zeros((VS, G, 4), dtype=np.int64)
        mean = np.zeros((VS, G, 4))
        star[sel] = self.haplo_SNP.tau_star
        star[~sel] = haplo_SNP_NS.tau_star
        mean[sel] = self.haplo_SNP.probabilisticTau()
        mean[~sel] = haplo_SNP_NS.probabilisticTau()
        names = full_variants.index.tolist()
        pos = full_variants['Position']
        orig = self.variantFilter.selected_indices_original
        o_names = [names[i] for i in orig]
        o_pos = [pos.iloc[i] for i in orig]
        self._position_first(star.reshape(VS, G * 4), o_names, o_pos).to_csv(self.outputDir + "/Collated_Tau_star.csv")
        logging.info("Wrote all tau haplotype predictions")
        self._position_first(mean.reshape(VS, G * 4), o_names, o_pos).to_csv(self.outputDir + "/Collated_Tau_mean.csv")
        logging.info("Wrote all probabilistic tau haplotype predictions")

    def _sample_names(self):
        cols = self.variants.columns.values.tolist()
        n0 = (len(cols) - 1) // 4
        names = [rchop(cols[i], '-A') for i in range(1, n0 * 4, 4)]
        return [names[i] for i in self.variantFilter.sample_indices]

    def output_Gamma_Mean(self, gamma):
        p.DataFrame(gamma, index=self._sample_names()).to_csv(self.outputDir + "/Gamma_mean.csv")
        logging.info("Wrote mean gamma haplotype relative frequencies")

    def output_Gamma(self, gamma):
        p.DataFrame(gamma, index=self._sample_names()).to_csv(self.outputDir + "/Gamma_star.csv")
        logging.info("Wrote gamma haplotype relative frequencies")

    def output_Eta(self, eta):
        p.DataFrame(eta).to_csv(self.outputDir + "/Eta_star.csv")
        logging.info("Wrote transition error matrix")

    def output_Eta_Mean(self, eta):
        p.DataFrame(eta).to_csv(self.outputDir + "/Eta_mean.csv")
        logging.info("Wrote transition error matrix")

    def output_Selected_Variants(self):
        self.variants[self.variantFilter.selected].to_csv(self.outputDir + "/Selected_variants.csv")
        logging.info("Wrote selected variants")
