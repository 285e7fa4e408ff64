"""Library builds of the NMF start side by side: us per update (wall + per launch) and whether traces / factors are the baseline's bits.
usage: nmft_variants.py name=path[,name=path...] "V S G[,fix]" ...      (each build runs in its own process: DESMAN_HIP_LIB)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, time, json, hashlib; sys.path.insert(0, %r)
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
V, S, G, fix, n = [int(x) for x in sys.argv[1:6]]
counts, _, _ = synth_counts(V, S, G, 1234)
ctx = _lib.Context(0); ctx.set_counts(counts)
rs = np.random.RandomState(0)
gam0 = np.ascontiguousarray(rs.dirichlet(np.full(G, 0.01), size=S).T)
d = rs.dirichlet(np.full(4, 0.01), size=V * G).reshape(V, G, 4)
tau0 = np.ascontiguousarray(np.transpose(d, (2, 0, 1)).reshape(4 * V, G))
if fix: gam0 = np.ascontiguousarray(rs.dirichlet(np.full(G, 1.0), size=S).T)
ctx.nmft_set(tau0, gam0); ctx.nmft_factorize(max_iter=5, min_change=0.0, fix_gamma=bool(fix))
ctx.nmft_set(tau0, gam0)
t0 = time.perf_counter(); nd, tr = ctx.nmft_factorize(max_iter=n, min_change=0.0, fix_gamma=bool(fix)); dt = time.perf_counter() - t0
t, g = ctx.nmft_get()
h = hashlib.sha1(np.ascontiguousarray(tr).tobytes() + t.tobytes() + g.tobytes()).hexdigest()[:12]
import os
ref = os.environ.get("NMV_REF")
dev = None
if ref and os.path.exists(ref):
    z = np.load(ref)
    rel = lambda a, b: float(np.max(np.abs(a - b) / (np.abs(b) + 1e-300)))
    dev = dict(tr=rel(np.asarray(tr), z["tr"]), tau=float(np.max(np.abs(t - z["t"]))), gam=float(np.max(np.abs(g - z["g"]))))
elif ref:
    np.savez(ref, tr=np.asarray(tr), t=t, g=g)
ctx.set_timing(True); ctx.nmft_factorize(max_iter=50, min_change=0.0, fix_gamma=bool(fix)); tm = ctx.get_timing(); ctx.set_timing(False)
print(json.dumps(dict(us=1e6 * dt / max(nd, 1), n=int(nd), div=float(tr[-1]), sha=h, dev=dev, k={k: round(1e3 * ms / max(c, 1), 1) for k, (ms, c) in tm.items() if c and k.startswith("nmft")})))
''' % ROOT
libs = [x.split("=") for x in sys.argv[1].split(",")]
for shp in sys.argv[2:]:
    a = shp.replace(",", " ").split()
    V, S, G = a[:3]; fix = a[3] if len(a) > 3 else "0"; n = a[4] if len(a) > 4 else "200"
    base = None
    ref = "/tmp/nmv_ref_%s_%s_%s_%s.npz" % (V, S, G, fix)
    if os.path.exists(ref): os.unlink(ref)
    for rep in range(2):
        for name, path in libs:
            env = dict(os.environ, DESMAN_HIP_LIB=os.path.join(ROOT, path), NMV_REF=ref)
            r = subprocess.run([sys.executable, "-c", CHILD, V, S, G, fix, n], env=env, capture_output=True, text=True)
            try:
                o = json.loads(r.stdout.strip().split("\n")[-1])
            except Exception:
                print("%s V=%s S=%s G=%s fix=%s: FAILED %s" % (name, V, S, G, fix, (r.stderr or r.stdout)[-300:])); continue
            if base is None: base = o["sha"]
            print("%-8s V=%s S=%s G=%s fix=%s: %.1f us/update  kernels %s  div %.9g  %s" % (name, V, S, G, fix, o["us"], o["k"], o["div"], "same bits as the first" if o["sha"] == base else "other bits: trace rel %.1e, tau abs %.1e, gamma abs %.1e" % (o["dev"]["tr"], o["dev"]["tau"], o["dev"]["gam"]) if o.get("dev") else "OTHER BITS"), flush=True)
