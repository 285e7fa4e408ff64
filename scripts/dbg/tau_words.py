"""How many DISTINCT tau words (rows of the haplotype table) a fitted chain holds -- what the word-pooled mu/E pass (spec 4) would pool over.
A `desman -g G -i I` chain on the config-5 table (50k x 96, six strains), then the unique rows of Filtered_Tau_star.csv.  usage: tau_words.py [G ...]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, pandas as p
from desman_amd import cli
from desman_amd.synth import synth_counts
V, S = 50000, 96
counts, _, _ = synth_counts(V, S, 6, seed=1234)
cols = ["Position"] + ["S%d-%s" % (s, b) for s in range(S) for b in "ACGT"]
df = p.DataFrame(np.concatenate([np.arange(V)[:, None] * 7 + 3, counts.reshape(V, S * 4)], axis=1), index=["contig%d" % (v // 50) for v in range(V)], columns=cols)
with tempfile.TemporaryDirectory() as d:
    freq = os.path.join(d, "syn.freq"); df.to_csv(freq)
    for G in [int(x) for x in sys.argv[1:]] or [8, 10, 12]:
        out = os.path.join(d, "o%d" % G)
        cli.main([freq, "-g", str(G), "-i", "200", "-o", out])
        t = p.read_csv(os.path.join(out, "Filtered_Tau_star.csv"), index_col=0).to_numpy()[:, 1:]
        u = np.unique(t, axis=0)
        print("G = %d: %d positions, %d distinct tau words (%.1f %%)" % (G, t.shape[0], u.shape[0], 100.0 * u.shape[0] / t.shape[0]), flush=True)
