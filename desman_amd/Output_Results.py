"""Result files of a `desman` run -- the on-disk half of the drop-in contract.

Same class / method names as desman/Output_Results.py (the driver and downstream scripts call them) and
byte-identical files (tests/test_host_cpu.py compares against files the reference class wrote); inside,
everything goes through two small helpers: one for haplotype tables (Position first, then 4 columns per
haplotype), one for per-sample abundance tables.
"""
import hashlib
import logging
import os
import shutil
import sys
import threading
import weakref

import numpy as np
import pandas as pd

_SELECTED_LOCK = threading.Lock()
_SELECTED_CACHE = {}        # (id of the table, rows, digest of the selection) -> (weak ref to the table, a file holding the text, its size)

LOG_FORMAT = '%(asctime)s:%(levelname)s:%(name)s:%(message)s'


def _table_bytes(flat, names, positions):
    """The text pandas' ``DataFrame.to_csv`` writes for a haplotype table (index = names, a Position column, then the columns
    0..n-1 of ``flat``), assembled without a Python-level loop over cells -- or None where that text is not plain enough to be
    assembled this way (the caller then lets pandas write it).  A table of a chain has few distinct values (0 / 1, or k / n_iter),
    so each distinct value is formatted once, exactly as pandas formats it (numpy's shortest round-trip text), and the cells are
    gathered as fixed-width byte fields whose padding is squeezed out at the end: V = 50k, G = 12 in 0.3 s instead of 2.9 s
    (Tau_Mean.csv) -- next to a GPU fit of about 2 s, the result files were a third of a chain's wall time."""
    flat = np.asarray(flat)
    pos = np.asarray(positions)
    if flat.ndim != 2 or flat.dtype.kind not in "iuf" or pos.dtype.kind not in "iu" or flat.shape[0] != len(names) or not flat.size:
        return None
    if flat.dtype.kind == "f" and (flat.dtype != np.float64 or not np.isfinite(flat).all()):
        return None
    if not all(type(n) is str and n and n.isascii() and not any(c in n for c in ',"\r\n\0') for n in names):
        return None                                       # (a non-ASCII contig name has no 'S' form: pandas writes it)
    n, nc = flat.shape
    if flat.dtype.kind == "f":                            # by bit pattern: 0.0 and -0.0 print differently
        inv, uniq = pd.factorize(np.ascontiguousarray(flat, dtype=np.float64).ravel().view(np.int64))
        uniq = np.asarray(uniq).view(np.float64)
    else:
        inv, uniq = pd.factorize(flat.ravel())
    if len(uniq) > 65536:
        return None
    as_bytes = lambda a: np.array(np.asarray(a).astype(str).tolist(), dtype="S")     # minimal field width
    ustr, nm, ps = as_bytes(uniq), np.array(list(names), dtype="S"), as_bytes(pos)
    k, kn, kp = ustr.dtype.itemsize, nm.dtype.itemsize, ps.dtype.itemsize
    row = np.zeros((n, kn + 1 + kp + 1 + nc * (k + 1)), dtype=np.uint8)
    row[:, :kn] = nm.view(np.uint8).reshape(n, kn)
    row[:, kn] = ord(",")
    row[:, kn + 1:kn + 1 + kp] = ps.view(np.uint8).reshape(n, kp)
    row[:, kn + 1 + kp] = ord(",")
    cells = row[:, kn + kp + 2:].reshape(n, nc, k + 1)
    cells[:, :, :k] = ustr[inv.reshape(n, nc)].view(np.uint8).reshape(n, nc, k)
    cells[:, :, k] = ord(",")
    cells[:, -1, k] = ord("\n")
    header = (",Position," + ",".join(str(i) for i in range(nc)) + "\n").encode()
    return header + row[row != 0].tobytes()


def rchop(text, suffix):
    return text[:-len(suffix)] if text.endswith(suffix) else text


class Output_Results:

    def __init__(self, outputDir):
        self.outputDir = outputDir
        os.makedirs(outputDir, exist_ok=True)
        self.log_file_name = outputDir + "/log_file.txt"
        logging.basicConfig(filename=self.log_file_name, level=logging.INFO, filemode='w', format=LOG_FORMAT)
        here = os.path.abspath(self.outputDir)
        logging.info("Results created in {0}".format(here))
        print("Up and running. Check {0} for progress".format(os.path.abspath(self.log_file_name)), file=sys.stderr)

    # ---- wiring
    def set_Variants(self, variants):
        self.variants = variants
        self.contig_names = variants.index.tolist()
        self.position = variants['Position']

    def set_Variant_Filter(self, variantFilter):
        self.variantFilter = variantFilter
        keep = variantFilter.selected_indices
        self.filtered_contig_names = [self.contig_names[i] for i in keep]
        self.filtered_position = self.position.to_numpy()[np.asarray(keep, dtype=np.int64)]

    def _path(self, name):
        return self.outputDir + "/" + name

    # ---- fit statistics: Fit,<G requested>,<G kept>,<lp*>,<mean deviance>
    def _write_fit(self, name, sampler, requested):
        line = "Fit,%d,%d,%f,%f\n" % (requested, sampler.G, sampler.lp_star, sampler.meanDeviance())
        with open(self._path(name), "w") as fh:
            fh.write(line)

    def set_haplo_SNP(self, haplo_SNP, genomes):
        self.haplo_SNP = haplo_SNP
        self._write_fit("fit.txt", haplo_SNP, genomes)
        logging.info("fit.txt written")

    def outPredFit(self, haplo_SNP, genomes):
        self._write_fit("fitP.txt", haplo_SNP, genomes)
        logging.info("fitP.txt written")

    # ---- haplotype tables
    def _haplotype_table(self, name, values, names, positions):
        """rows = positions, first column Position, then the flattened [G][4] block."""
        flat = np.reshape(values, (values.shape[0], -1))
        text = _table_bytes(flat, names, positions)
        if text is not None:
            with open(self._path(name), "wb") as fh:
                fh.write(text)
            return
        frame = pd.DataFrame(flat, index=names)
        frame['Position'] = positions
        order = frame.columns.tolist()
        frame[order[-1:] + order[:-1]].to_csv(self._path(name))

    def output_Filtered_Tau(self, tau):
        self._haplotype_table("Filtered_Tau_star.csv", np.reshape(tau, (self.haplo_SNP.V, self.haplo_SNP.G, 4)),
                              self.filtered_contig_names, self.filtered_position)
        logging.info("Filtered_Tau_star.csv written")

    def output_Tau_Mean(self, tauProb):
        self._haplotype_table("Tau_Mean.csv", np.reshape(tauProb, (self.haplo_SNP.V, self.haplo_SNP.G, 4)),
                              self.filtered_contig_names, self.filtered_position)
        logging.info("Tau_Mean.csv written")

    def output_collated_Tau(self, haplo_SNP_NS, full_variants):
        """-r: fitted and assigned positions merged back into input order."""
        total = haplo_SNP_NS.V + self.haplo_SNP.V
        G = self.haplo_SNP.G
        fitted = np.asarray(self.variantFilter.selected[:total], dtype=bool)
        star = np.zeros((total, G, 4), dtype=np.int64)
        prob = np.zeros((total, G, 4))
        star[fitted], star[~fitted] = self.haplo_SNP.tau_star, haplo_SNP_NS.tau_star
        prob[fitted], prob[~fitted] = self.haplo_SNP.probabilisticTau(), haplo_SNP_NS.probabilisticTau()
        all_names = full_variants.index.tolist()
        all_pos = full_variants['Position']
        rows = self.variantFilter.selected_indices_original
        names = [all_names[i] for i in rows]
        positions = all_pos.to_numpy()[np.asarray(rows, dtype=np.int64)]
        self._haplotype_table("Collated_Tau_star.csv", star, names, positions)
        self._haplotype_table("Collated_Tau_mean.csv", prob, names, positions)
        logging.info("Collated_Tau_star.csv / Collated_Tau_mean.csv written")

    # ---- abundance and error tables
    def _kept_sample_names(self):
        cols = self.variants.columns.values.tolist()
        n_samples = (len(cols) - 1) // 4
        every = [rchop(cols[1 + 4 * k], '-A') for k in range(n_samples)]
        return [every[k] for k in self.variantFilter.sample_indices]

    def _abundance_table(self, name, gamma):
        pd.DataFrame(gamma, index=self._kept_sample_names()).to_csv(self._path(name))
        logging.info(name + " written")

    def output_Gamma(self, gamma):
        self._abundance_table("Gamma_star.csv", gamma)

    def output_Gamma_Mean(self, gamma):
        self._abundance_table("Gamma_mean.csv", gamma)

    def output_Eta(self, eta):
        pd.DataFrame(eta).to_csv(self._path("Eta_star.csv"))
        logging.info("Eta_star.csv written")

    def output_Eta_Mean(self, eta):
        pd.DataFrame(eta).to_csv(self._path("Eta_mean.csv"))
        logging.info("Eta_mean.csv written")

    def output_Selected_Variants(self):
        """the selected rows of the input table.  Every chain of a G-sweep writes the same file (the -r selection is drawn from
        a fixed seed, bin/desman:85-86): serialising 50 000 x 385 integers takes longer than the chain's GPU work, so within one
        process the text is produced once per (table, selection) and copied afterwards."""
        path = self._path("Selected_variants.csv")
        sel = np.asarray(self.variantFilter.selected, dtype=bool)
        key = (id(self.variants), sel.shape[0], hashlib.blake2b(np.packbits(sel).tobytes(), digest_size=16).digest())
        with _SELECTED_LOCK:
            prev = _SELECTED_CACHE.get(key)
            if prev is not None and prev[0]() is self.variants and os.path.isfile(prev[1]) and os.path.getsize(prev[1]) == prev[2] \
                    and os.path.abspath(prev[1]) != os.path.abspath(path):
                shutil.copyfile(prev[1], path)
            else:
                self.variants[sel].to_csv(path)
                while len(_SELECTED_CACHE) >= 4:
                    _SELECTED_CACHE.pop(next(iter(_SELECTED_CACHE)))
                _SELECTED_CACHE[key] = (weakref.ref(self.variants), path, os.path.getsize(path))
        logging.info("Selected_variants.csv written")
