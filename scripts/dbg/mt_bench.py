"""MT19937 generator: us per fill by the library's own event timers (set_timing), serial vs from several CUs.
usage: mt_bench.py          (DESMAN_HIP_LIB=.../libdesman_hip_ab.so DESMAN_HIP_MT_SERIAL=1 for the serial generator)"""
import sys, time; sys.path.insert(0, '.')
import numpy as np
from desman_amd import _lib
ctx = _lib.Context(0); ctx.seed(1)
ctx.debug_mt_fill(8 * 131040)            # jump tables built here
for n in (80000, 400000, 6 * 131040, 1200000, 3200000, 10 ** 7):
    ctx.debug_mt_fill(n)
    ctx.set_timing(True)
    for _ in range(5): ctx.debug_mt_fill(n)
    tm = ctx.get_timing(); ctx.set_timing(False)
    ms, cnt = tm["mt"]
    print("n = %8d words: %.1f us per fill (%.2f words/ns)" % (n, 1e3 * ms / cnt, n * cnt / ms * 1e-6), flush=True)
