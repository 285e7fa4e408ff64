"""Posterior-deviance model selection over a G-sweep (row f2 of SURVEY sec. 8):
the heuristic of the reference's scripts/resolvenhap.py, operating on the same
`<stub>_<G>_<r>/` output directories and writing the same `*R.csv` files.

Rule (resolvenhap.py:118-217): mean posterior deviance per G over the
replicates that kept all G haplotypes; walk G upwards while the fractional
reduction stays >= delta_g; for every retained G take the lowest-deviance
replicate, score each haplotype by its mean SNV disagreement with its greedy
best match in the other replicates, count haplotypes with mean abundance >
min_freq and error < max_err; choose the G with most such haplotypes (ties:
lower mean error, then smaller G).
"""
import argparse
import glob
import os
import re
import sys
from collections import defaultdict

import numpy as np
import pandas as p


def comp_snd(tau1, tau2):
    """pairwise Hamming distance between the haplotypes of two runs (resolvenhap.py:36-54)."""
    i1, i2 = np.argmax(tau1, axis=2), np.argmax(tau2, axis=2)
    return (i1[:, :, None] != i2[:, None, :]).sum(axis=0)


def _read_tau(path):
    m = p.read_csv(path, header=0, index_col=0).to_numpy()[:, 1:]
    V, G = m.shape[0], m.shape[1] // 4
    return np.reshape(m, (V, G, 4))


def strain_reproducibility(gamma_file, tau_file, comp_files):
    """(mean abundance per haplotype, mean matched SNV error per haplotype) (resolvenhap.py:57-117)."""
    gamma_mean = np.mean(p.read_csv(gamma_file, header=0, index_col=0).to_numpy(), axis=0)
    tau = _read_tau(tau_file)
    V, G = tau.shape[0], tau.shape[1]
    acc = np.zeros((G, len(comp_files)))
    for c, f in enumerate(comp_files):
        other = _read_tau(f)
        if other.shape[0] != V or other.shape[1] != G:
            print('Haplotype files do not match V %d -> %d or G %d -> %d' % (V, other.shape[0], G, other.shape[1]))
            sys.exit(-1)
        comp = comp_snd(tau, other) / float(V)
        for _ in range(G):                                   # greedy one-to-one matching
            r, col = np.unravel_index(np.argmin(comp), comp.shape)
            acc[r, c] = comp[r, col]
            comp[r, :] = 1.0
            comp[:, col] = 1.0
    mean_acc = np.mean(acc, axis=1) if len(comp_files) > 0 else np.ones(G)
    return gamma_mean, mean_acc


def resolve(input_stub, delta_g=0.05, max_err=0.10, min_freq=0.05, write=True):
    """Returns (bestG, NStrains, best replicate, mean error, tau file, selected strains) or None."""
    g_values = []
    for d in glob.glob(input_stub + "_*_0"):
        if os.path.isdir(d):
            m = re.match(r'.*_(\d+)_0', d)
            if m:
                g_values.append(int(m.group(1)))
    g_values = sorted(g_values)
    NG = len(g_values)
    sum_pd, count_pd = np.zeros(NG), np.zeros(NG, dtype=int)
    all_pd = defaultdict(dict)
    gidx = 0
    for G in g_values:
        for fit_file in glob.glob(input_stub + "_" + str(G) + "_*/fit.txt"):
            r = int(re.match(".*_" + str(G) + r"_(\d+)", fit_file).group(1))
            with open(fit_file) as f:
                _, GT, HT, LL, PD = f.readline().strip().split(',')
            if int(HT) == G:                                 # replicates that lost a haplotype are not used
                all_pd[G][r] = float(PD)
                sum_pd[gidx] += float(PD)
                count_pd[gidx] += 1
        gidx += 1
    with np.errstate(divide='ignore', invalid='ignore'):
        mean_pd = sum_pd / count_pd
    # NB the reference re-uses `gidx` after this loop (resolvenhap.py:184-190): with fewer than three G
    # values the loop body never runs and every G is kept; when it runs to completion the last G is dropped.
    for gidx in range(2, NG):
        frac = (mean_pd[gidx - 1] - mean_pd[gidx]) / mean_pd[gidx - 1]
        if count_pd[gidx] < 1 or frac < delta_g:
            break
    new_ng = gidx
    quality = {}
    for k in range(new_ng):
        G = g_values[k]
        if len(all_pd[G]) > 0:
            bestr = min(all_pd[G], key=all_pd[G].get)
            d = input_stub + "_" + str(G) + "_"
            comp_files = [d + str(r) + "/Filtered_Tau_star.csv" for r in all_pd[G] if r != bestr]
            gamma_mean, mean_acc = strain_reproducibility(d + str(bestr) + "/Gamma_star.csv",
                                                          d + str(bestr) + "/Filtered_Tau_star.csv", comp_files)
            sel = [h for h, (m, a) in enumerate(zip(gamma_mean, mean_acc)) if m > min_freq and a < max_err]
            if sel:
                mean_error = float(np.mean([mean_acc[h] for h in sel]))
            else:
                best = int(np.argmin(mean_acc))
                sel, mean_error = [best], mean_acc[best]
            quality[G] = (len(sel), mean_error, bestr, sel, G)
        else:
            quality[G] = (0, 1.0, -1, None, G)
    if new_ng <= 0 or not quality:
        return None
    order = sorted(quality, key=lambda k: (quality[k][0], -quality[k][1], -quality[k][4]))
    bestG = order[-1]
    n_strains, mean_error, bestr, sel, _ = quality[bestG]
    d = input_stub + "_" + str(bestG) + "_" + str(bestr) + "/"
    tau_file = d + "Filtered_Tau_star.csv"
    print(str(bestG) + "," + str(n_strains) + "," + str(bestr) + "," + str(mean_error) + "," + tau_file)
    if write:
        cols = [0] + [1 + 4 * h + n for h in sel for n in range(4)]
        for name in ("Gamma_star", "Gamma_mean"):
            p.read_csv(d + name + ".csv", header=0, index_col=0).iloc[:, sel].to_csv(d + name + "R.csv")
        for name, out in (("Filtered_Tau_star", "Filtered_Tau_starR"), ("Tau_Mean", "Tau_MeanR")):
            p.read_csv(d + name + ".csv", header=0, index_col=0).iloc[:, cols].to_csv(d + out + ".csv")
        if os.path.isfile(d + "Collated_Tau_star.csv"):
            for name, out in (("Collated_Tau_star", "Collated_Tau_starR"), ("Collated_Tau_mean", "Collated_Tau_meanR")):
                p.read_csv(d + name + ".csv", header=0, index_col=0).iloc[:, cols].to_csv(d + out + ".csv")
    return bestG, n_strains, bestr, mean_error, tau_file, sel


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("input_stub", help="relative frequencies of haplotypes")
    ap.add_argument('-d', '--delta_g', type=float, default=0.05, help="minimum fractional reduction in PD default 0.05")
    ap.add_argument('-m', '--max_err', type=float, default=0.10, help="maximum error valid strain")
    ap.add_argument('-f', '--min_freq', type=float, default=0.05, help="minimum frequency valid strain")
    a = ap.parse_args(argv)
    resolve(a.input_stub, a.delta_g, a.max_err, a.min_freq)


if __name__ == "__main__":
    main(sys.argv[1:])
