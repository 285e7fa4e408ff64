"""round 6: the NMF start of a batch of small tables -- every chain its own persistent launch, as many at once as fit -- against the
batched three-launch loop (DESMAN_HIP_NMFT_NO_PERSIST=1, experiment build) and against the chains one by one.  us per chain-update.  usage: r06_batch_nmft.py"""
import os, sys, subprocess, time
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = os.path.join(root, "desman_amd", "lib", "libdesman_hip_ab.so")
code = r'''
import sys, time; sys.path.insert(0, %r)
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
shape, K, fix, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
if shape == "cog":
    counts = np.ascontiguousarray(np.load(%r)["counts"].astype(np.int64)); V, S = counts.shape[:2]; G = 5
else:
    V, S, G = (int(x) for x in shape.split("x")); counts, _, _ = synth_counts(V, S, min(G, 4), seed=3)
rs = np.random.RandomState(1)
ctxs = []
for k in range(K):
    gam0 = np.ascontiguousarray(rs.dirichlet(np.full(G, 0.01), size=S).T)
    d = rs.dirichlet(np.full(4, 0.01), size=V * G).reshape(V, G, 4)
    tau0 = np.ascontiguousarray(np.transpose(d, (2, 0, 1)).reshape(4 * V, G))
    c = _lib.Context(0); c.set_counts(counts); c.nmft_set(tau0, gam0); ctxs.append((c, tau0, gam0))
N = 2000
def run():
    for c, t, g in ctxs: c.nmft_set(t, g)
    t0 = time.perf_counter()
    if mode == "batch":
        res = _lib.Context.batch_nmft_factorize([c for c, _, _ in ctxs], N, 0.0, fix_gamma=bool(fix))
        n = sum(r[0] for r in res)
    else:
        n = 0
        for c, _, _ in ctxs: n += c.nmft_factorize(N, 0.0, fix_gamma=bool(fix))[0]
    return (time.perf_counter() - t0) / n * 1e6
run()
print("%%.2f" %% min(run() for _ in range(3)))
''' % (root, os.path.join(root, "tests", "golden", "cog0015_counts.npz"))
def run(shape, K, fix, mode, **env):
    e = dict(os.environ, DESMAN_HIP_LIB=lib, **env)
    r = subprocess.run([sys.executable, "-c", code, shape, str(K), str(fix), mode], env=e, capture_output=True, text=True)
    return r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "ERR " + r.stderr[-200:]
for shape in ("1000x64x5", "cog", "1000x16x5", "3000x64x8", "10000x64x8"):
    for fix in (0, 1):
        print("%-10s K=8 %s: one by one %s | dsm_batch_nmft_factorize %s %s   (us per chain-update, best of 3, twice)" % (
            shape, "factorize_tau" if fix else "factorize    ", run(shape, 8, fix, "single"), run(shape, 8, fix, "batch"), run(shape, 8, fix, "batch")), flush=True)
