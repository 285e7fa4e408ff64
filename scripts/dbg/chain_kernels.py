"""one whole `desman` chain at the config-5 shape (table from six strains) -- run under rocprofv3 --kernel-trace --stats to see which
kernels a REAL chain spends its time in (the state after 5000 NMF updates and the burn-in is not bench.py's).  usage: chain_kernels.py G"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, pandas as p
from desman_amd import cli
from desman_amd.synth import synth_counts
V, S, G = 50000, 96, int(sys.argv[1])
counts, _, _ = synth_counts(V, S, 6, seed=1234)
cols = ["Position"] + ["S%d-%s" % (s, b) for s in range(S) for b in "ACGT"]
data = np.concatenate([np.arange(V)[:, None] * 7 + 3, counts.reshape(V, S * 4)], axis=1)
df = p.DataFrame(data, index=["contig%d" % (v // 50) for v in range(V)], columns=cols)
d = tempfile.mkdtemp()
freq = os.path.join(d, "syn.freq"); df.to_csv(freq)
cli.main([freq, "-g", str(G), "-i", "500", "-o", os.path.join(d, "out")])
print(open(os.path.join(d, "out", "fit.txt")).read())
