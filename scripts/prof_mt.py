"""Times the MT19937 fill kernel alone (no other work on the GPU): python scripts/prof_mt.py [n_words reps]"""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from desman_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 80000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ctx = _lib.Context(0); ctx.seed(1)
ctx.debug_mt_fill(n)
ctx.set_timing(True)
for _ in range(reps):
    ctx.debug_mt_fill(n)
tm = ctx.get_timing(); ctx.set_timing(False)
print(n, "words:", {k: round(1e3 * ms / max(c, 1), 1) for k, (ms, c) in tm.items() if c})
