"""round 6: phase stamps of the Dirichlet launch's rows (experiment build): stage 2's levels and the gamma / eta draws.  usage: r06_s2_clocks.py [V S G]"""
import os, sys, ctypes
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DESMAN_HIP_LIB", os.path.join(root, "desman_amd", "lib", "libdesman_hip_ab.so"))
sys.path.insert(0, root)
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
from oracle import cbind
V, S, G = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (10000, 64, 8)
counts, tt, gg = synth_counts(V, S, G, 1234)
ctx = _lib.Context(0); ctx.set_counts(counts); ctx.seed(0)
ctx.set_state(cbind.idx_to_onehot(tt), np.ascontiguousarray(gg), 0.96 * np.eye(4) + 0.01)
ctx.gibbs_update(60)
ctx.set_timing(True); ctx.gibbs_update(20); tm = ctx.get_timing(); ctx.set_timing(False)
print({k: round(1e3 * a / max(b, 1), 1) for k, (a, b) in tm.items() if b})
lib = _lib.load()
lib.dsm_debug_s2_clocks.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = np.zeros((1024, 8), np.uint64)
lib.dsm_debug_s2_clocks(buf.ctypes.data, 1024)
b = buf.astype(np.int64)
rows = b[:S + 4]
g = rows[:S]; e = rows[S:S + 4]
t0 = g[:, 0].min()
us = lambda x: (x - t0) / 100.0
q = lambda x: "min %.1f med %.1f max %.1f" % tuple(np.percentile(x, [0, 50, 100]))
names = ["entry", "tables staged", "level 0 (root: table reads + binomials)", "level 1", "level 2", "level 3", "stage 2 done", "row written"]
print("gamma rows (last iteration), us since the first row's entry:")
prev = None
for k, nm in enumerate(names):
    col = g[:, k]
    if (col > 0).all():
        print("  %-42s %s%s" % (nm, q(us(col)), ("   | since the previous stamp " + q((col - prev) / 100.0)) if prev is not None else ""))
        prev = col
print("eta rows: fold + draw: entry %s -> written %s" % (q(us(e[:, 6])), q(us(e[:, 7]))))
