"""round 6: the time line of stage 1's wavefronts (experiment build: s_memrealtime stamps, 10 ns units).  usage: r06_s1_clocks.py [V S G] [dbg]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DESMAN_HIP_LIB", os.path.join(root, "desman_amd", "lib", "libdesman_hip_ab.so"))
os.environ.setdefault("DESMAN_HIP_NTAB_OFF", "256")
sys.path.insert(0, root)
import ctypes
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
from oracle import cbind
V, S, G = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (10000, 64, 8)
counts, tt, gg = synth_counts(V, S, G, 1234)
ctx = _lib.Context(0); ctx.set_counts(counts); ctx.seed(0)
ctx.set_state(cbind.idx_to_onehot(tt), np.ascontiguousarray(gg), 0.96 * np.eye(4) + 0.01)
ctx.force_stats_spec(2)
for it in range(20): ctx.sample_stats(it)
lib = _lib.load()
lib.dsm_debug_s1_clocks.argtypes = [ctypes.c_void_p, ctypes.c_int]
ctx.sample_stats(999)
ctx.sync() if hasattr(ctx, "sync") else None
buf = np.zeros((8192, 16), np.uint64)
n = lib.dsm_debug_s1_clocks(buf.ctypes.data, 8192)
b = buf.astype(np.int64)
live = b[:, 0] > 0
b = b[live]
t0 = b[:, 0].min()
us = lambda x: (x - t0) / 100.0
print("waves", len(b), "kernel span %.1f us (first entry -> last end)" % us(b[:, 15].max()))
def q(x): return "min %.1f p10 %.1f med %.1f p90 %.1f max %.1f" % tuple(np.percentile(x, [0, 10, 50, 90, 100]))
print("entry          :", q(us(b[:, 0])))
print("tables staged  :", q(us(b[:, 1])), "| prologue dt:", q((b[:, 1] - b[:, 0]) / 100.0))
for p_ in range(3):
    k = 2 + 4 * p_
    m = b[:, k + 3] > 0
    if not m.any(): break
    bb = b[m]
    print("pass %d (%d waves): start %s" % (p_ + 1, m.sum(), q(us(bb[:, k]))))
    print("     load+Gamma+Philox dt:", q((bb[:, k + 1] - bb[:, k]) / 100.0))
    print("     four items        dt:", q((bb[:, k + 2] - bb[:, k + 1]) / 100.0))
    print("     table atomics     dt:", q((bb[:, k + 3] - bb[:, k + 2]) / 100.0))
    print("     end  :", q(us(bb[:, k + 3])))
print("before epilogue:", q(us(b[:, 14])))
print("end            :", q(us(b[:, 15])), "| epilogue dt:", q((b[:, 15] - b[:, 14]) / 100.0))

# where the wavefronts ran: per CU (XCC, SE, SH, CU id), how many wavefronts and passes, when its last wavefront left the pass loop
hw = (b[:, 13] >> 8) & 0xFFFFFFFF
xcc = b[:, 13] & 15
cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
passes = (b[:, 5] > 0).astype(int) + (b[:, 9] > 0).astype(int) + (b[:, 13 - 0] * 0).astype(int)
passes = (b[:, 2 + 3] > 0).astype(int) + (b[:, 6 + 3] > 0).astype(int) + (b[:, 10 + 3] > 0).astype(int)
import collections
per = collections.defaultdict(lambda: [0, 0, 0.0])
for k_, p2, e in zip(key, passes, us(b[:, 14])):
    r = per[int(k_)]; r[0] += 1; r[1] += int(p2); r[2] = max(r[2], float(e))
nw = np.array([r[0] for r in per.values()]); npass = np.array([r[1] for r in per.values()]); tend = np.array([r[2] for r in per.values()])
print("CUs seen", len(per), "| wavefronts per CU:", dict(collections.Counter(nw.tolist())), "| passes per CU:", q(npass))
for lo, hi in ((0, 30), (30, 34), (34, 38), (38, 42), (42, 46), (46, 99)):
    m = (npass >= lo) & (npass < hi)
    if m.any(): print("  CUs with %d-%d passes: %d, last wavefront out of the pass loop at %s" % (lo, hi - 1, m.sum(), q(tend[m])))
print("corr(passes per CU, end) = %.3f" % np.corrcoef(npass, tend)[0, 1])
