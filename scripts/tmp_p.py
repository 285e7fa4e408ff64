import sys, os, cProfile, pstats, tempfile, io
sys.path.insert(0, '.')
import numpy as np, pandas as p
from desman_amd import cli
from desman_amd.synth import synth_counts
V, S = 1000, 32
counts, _, _ = synth_counts(V, S, 4, seed=7)
cols = ["Position"] + ["S%d-%s" % (s, b) for s in range(S) for b in "ACGT"]
data = np.concatenate([np.arange(V)[:, None] * 7 + 3, counts.reshape(V, S * 4)], axis=1)
df = p.DataFrame(data, index=["contig%d" % (v // 50) for v in range(V)], columns=cols)
with tempfile.TemporaryDirectory() as d:
    freq = os.path.join(d, "syn.freq"); df.to_csv(freq)
    cli.main([freq, "-g", "4", "-s", "0", "-i", "100", "-o", os.path.join(d, "w")])      # warm
    pr = cProfile.Profile(); pr.enable()
    cli.main([freq, "-g", "4", "-s", "1", "-i", "100", "-o", os.path.join(d, "x")])
    pr.disable()
    st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(35); print(st.getvalue()[:6000])
