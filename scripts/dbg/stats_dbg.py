"""timing experiments on stage 1 of the mu/E pass: the launch with parts of the work removed (results are garbage then)"""
import os as _os; _os.environ.setdefault("DESMAN_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))), "desman_amd", "lib", "libdesman_hip_ab.so"))  # the experiment build: A/B switches compiled in (make -C desman_amd/csrc ab)
import os, sys, subprocess, json
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys; sys.path.insert(0, %r)
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
V, S, G = 10000, 64, 8
counts, tt, gg = synth_counts(V, S, G, 1234)
from oracle import cbind
ctx = _lib.Context(0); ctx.set_counts(counts); ctx.seed(0)
ctx.set_state(cbind.idx_to_onehot(tt), np.ascontiguousarray(gg), 0.96 * np.eye(4) + 0.01)
ctx.force_stats_spec(int(sys.argv[1]))
for it in range(20): ctx.sample_stats(it)
ctx.set_timing(True)
for it in range(200): ctx.sample_stats(100 + it)
tm = ctx.get_timing()
print({k: round(1e3 * ms / max(n, 1), 1) for k, (ms, n) in tm.items() if n})
''' % root
for spec, xcd in ((2, 1), (3, 1)):
    for dbg in (0, 1, 2, 8, 10, 16, 32, 48, 59):
        env = dict(os.environ, DESMAN_HIP_STATS_DBG=str(dbg), DESMAN_HIP_STATS_REGG="0", DESMAN_HIP_NTAB_XCD=str(xcd))
        r = subprocess.run([sys.executable, "-c", code, str(spec)], env=env, capture_output=True, text=True)
        print("spec", spec, "xcd", xcd, "dbg", dbg, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
