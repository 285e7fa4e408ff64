"""round 6: stage 1 at config 3 for table places 0 .. 1792 B and row-map swizzles (experiment build).  usage: r06_swz_scan.py [V S G]"""
import os, sys, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = os.path.join(root, "desman_amd", "lib", "libdesman_hip_ab.so")
V, S, G = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (10000, 64, 8)
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
ctx.set_timing(True)
for it in range(200): ctx.sample_stats(100 + it)
tm = ctx.get_timing()
print(round(1e3 * tm["stats"][0] / tm["stats"][1], 1), {k: round(1e3 * a / max(b, 1), 1) for k, (a, b) in tm.items() if b})
''' % (root, V, S, G)
def run(**env):
    e = dict(os.environ, DESMAN_HIP_LIB=lib, **{k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
    return r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
for swz in (0, 1, 3, 17, 64, 85):
    print("swz %3d:" % swz, "  ".join("+%d: %s" % (off, run(DESMAN_HIP_NTAB_SWZ=swz, DESMAN_HIP_NTAB_OFF=off).split()[0]) for off in (0, 256, 512, 768, 1024, 1280, 4096, 65536)), flush=True)
print("detail swz 0 +256:", run(DESMAN_HIP_NTAB_SWZ=0, DESMAN_HIP_NTAB_OFF=256))
print("detail swz 17 +256:", run(DESMAN_HIP_NTAB_SWZ=17, DESMAN_HIP_NTAB_OFF=256))
