"""cProfile of one whole `desman` chain at the config-5 shape (host side)"""
import cProfile, pstats, os, sys, tempfile, time, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, pandas as p
from desman_amd import cli
from desman_amd.synth import synth_counts
V, S, G = 50000, 96, 8
counts, _, _ = synth_counts(V, S, 6, seed=1234)
cols = ["Position"] + ["S%d-%s" % (s, b) for s in range(S) for b in "ACGT"]
data = np.concatenate([np.arange(V)[:, None] * 7 + 3, counts.reshape(V, S * 4)], axis=1)
df = p.DataFrame(data, index=["contig%d" % (v // 50) for v in range(V)], columns=cols)
with tempfile.TemporaryDirectory() as d:
    freq = os.path.join(d, "syn.freq"); df.to_csv(freq)
    cli.main([freq, "-g", "3", "-i", "20", "-o", os.path.join(d, "warm")])       # warm (library load, table cache)
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
    cli.main([freq, "-g", str(G), "-i", "500", "-o", os.path.join(d, "out")])
    pr.disable(); print("chain wall %.2f s" % (time.perf_counter() - t0))
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:5000])
