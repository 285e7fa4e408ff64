"""stage 1 of the mu/E pass against the place of the subset table, all offsets inside ONE process (one set of physical pages):
is the cost of an offset stable, and how far apart are the best and the worst?  usage: ntab_scan_inproc.py [V S G]"""
import os as _os; _os.environ.setdefault("DESMAN_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))), "desman_amd", "lib", "libdesman_hip_ab.so"))  # the experiment build: A/B switches compiled in (make -C desman_amd/csrc ab)
import os, sys
os.environ["DESMAN_HIP_NTAB_SCAN"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
from oracle import cbind
V, S, G = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (10000, 64, 8)
counts, tt, gg = synth_counts(V, S, G, 1234)
ctx = _lib.Context(0); ctx.set_counts(counts); ctx.seed(0)
ctx.set_state(cbind.idx_to_onehot(tt), np.ascontiguousarray(gg), 0.96 * np.eye(4) + 0.01)
ctx.force_stats_spec(2)
offs = list(range(0, 8192, 256))
res = {o: [] for o in offs}
for rep in range(3):
    for o in offs:
        os.environ["DESMAN_HIP_NTAB_OFF"] = str(o)
        for it in range(5): ctx.sample_stats(it)
        ctx.set_timing(True)
        for it in range(100): ctx.sample_stats(100 + it)
        tm = ctx.get_timing(); ctx.set_timing(False)
        ms, n = tm["stats"]
        res[o].append(round(1e3 * ms / n, 1))
for o in offs: print(o, res[o])
best = min(offs, key=lambda o: np.median(res[o])); worst = max(offs, key=lambda o: np.median(res[o]))
print("best", best, res[best], "worst", worst, res[worst], "median of medians", np.median([np.median(v) for v in res.values()]))
