"""round 6 diagnostics of stage 1 (experiment builds): ablations, workgroups per CU, table place fixed.  usage: r06_stats_diag.py lib.so [V S G]"""
import os, sys, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = os.path.abspath(sys.argv[1])
V, S, G = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (10000, 64, 8)
code = r'''
import sys; sys.path.insert(0, %r)
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
from oracle import cbind
V, S, G = %d, %d, %d
counts, tt, gg = synth_counts(V, S, G, 1234)
ctx = _lib.Context(0); ctx.set_counts(counts); ctx.seed(0)
ctx.set_state(cbind.idx_to_onehot(tt), np.ascontiguousarray(gg), 0.96 * np.eye(4) + 0.01)
ctx.force_stats_spec(2)
for it in range(20): ctx.sample_stats(it)
res = []
for rep in range(3):
    ctx.set_timing(True)
    for it in range(200): ctx.sample_stats(100 + it)
    tm = ctx.get_timing(); ctx.set_timing(False)
    res.append(round(1e3 * tm["stats"][0] / tm["stats"][1], 2))
print(res)
''' % (root, V, S, G)
def run(**env):
    e = dict(os.environ, DESMAN_HIP_LIB=lib, **{k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
    return r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
print("lib", os.path.basename(lib), (V, S, G), flush=True)
for off in (0, 256, 512, 768, 1024, 1280):
    print("place +%d:" % off, run(DESMAN_HIP_NTAB_OFF=off), flush=True)
print("default (measured place):", run(), flush=True)
OFF = 256
for dbg in (1, 2, 3, 8, 11, 20, 36, 52, 63):
    print("dbg %2d:" % dbg, run(DESMAN_HIP_STATS_DBG=dbg, DESMAN_HIP_NTAB_OFF=OFF), flush=True)
for w in (3, 4, 5, 6, 7, 8):
    print("wgs %d:" % w, run(DESMAN_HIP_STATS_WGS=w, DESMAN_HIP_NTAB_OFF=OFF), "leave0", run(DESMAN_HIP_STATS_WGS=w, DESMAN_HIP_NTAB_OFF=OFF, DESMAN_HIP_STATS_LEAVE_CUS=0), flush=True)
