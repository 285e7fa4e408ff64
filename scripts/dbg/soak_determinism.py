"""two fresh processes run the benchmark's chain (NMF start, 3000 Gibbs iterations in calls of 7, 500, 1 ... iterations) and must agree bit for bit:
the table place, the block order of the sweep, the persistent NMF launch and the screening rule differ from process to process in WHERE
and WHEN things run, never in what they compute"""
import hashlib, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, hashlib; sys.path.insert(0, %r)
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
V, S, G = 10000, 64, 8
counts, _, _ = synth_counts(V, S, G, 1234)
ctx = _lib.Context(0); ctx.set_counts(counts); ctx.seed(5)
rs = np.random.RandomState(0)
gam0 = np.ascontiguousarray(rs.dirichlet(np.full(G, 0.01), size=S).T)
d = rs.dirichlet(np.full(4, 0.01), size=V * G).reshape(V, G, 4)
tau0 = np.ascontiguousarray(np.transpose(d, (2, 0, 1)).reshape(4 * V, G))
ctx.nmft_set(tau0, gam0); ctx.nmft_factorize(max_iter=300, min_change=0.0)
t = ctx.nmft_get_tau(); _, g = ctx.nmft_get()
ctx.set_state(t, np.ascontiguousarray(g.T), 0.96 * np.eye(4) + 0.01)
h = hashlib.sha256()
for n in (7, 500, 1, 993, 20, 1479):
    ctx.gibbs_update(n)
    tr = ctx.get_trace()
    for k in ("ll", "lp", "nchange", "gamma", "eta"): h.update(np.ascontiguousarray(tr[k]).tobytes())
    tt, gg, ee = ctx.get_state(); h.update(tt.tobytes()); h.update(gg.tobytes()); h.update(ee.tobytes())
print(h.hexdigest(), tr["ll"][-1])
''' % root
outs = []
for env in ({}, {"DESMAN_HIP_NTAB_TUNE": "0", "DESMAN_HIP_TAU_ORDER": "0"}, {}):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-500:]); outs.append(r.stdout.split()[0] if r.stdout else None)
print("identical" if len(set(outs)) == 1 and outs[0] else "DIFFERENT")
