"""stage 1 of the mu/E pass against the offset of the subset table inside its 2 MB-aligned block (DESMAN_HIP_NTAB_OFF): us per launch
(hipEvent timing of `stats`), at config 3 by default.  usage: ntab_off_scan.py [V S G]"""
import os as _os; _os.environ.setdefault("DESMAN_HIP_LIB", _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))), "desman_amd", "lib", "libdesman_hip_ab.so"))  # the experiment build: A/B switches compiled in (make -C desman_amd/csrc ab)
import os, sys, subprocess
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
V, S, G = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (10000, 64, 8)
code = r'''
import sys; sys.path.insert(0, %r)
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
V, S, G = %d, %d, %d
counts, tt, gg = synth_counts(V, S, G, 1234)
from oracle import cbind
ctx = _lib.Context(0); ctx.set_counts(counts); ctx.seed(0)
ctx.set_state(cbind.idx_to_onehot(tt), np.ascontiguousarray(gg), 0.96 * np.eye(4) + 0.01)
ctx.force_stats_spec(2)
for it in range(20): ctx.sample_stats(it)
ctx.set_timing(True)
for it in range(300): ctx.sample_stats(100 + it)
tm = ctx.get_timing()
print({k: round(1e3 * ms / max(n, 1), 1) for k, (ms, n) in tm.items() if n})
''' % (root, V, S, G)
offs = [int(x) for x in os.environ["OFFS"].split(",")] if os.environ.get("OFFS") else list(range(0, 16384, 256)) + [32768, 65536 + 256]
for off in offs:
    env = dict(os.environ, DESMAN_HIP_NTAB_OFF=str(off))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("off %8d" % off, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
