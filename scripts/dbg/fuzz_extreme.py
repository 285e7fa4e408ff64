"""round 6: many seeds of the exponent-range legs of tests/test_gpu_fuzz.py (tau sweep through the shim, LRT step, KL assignment): hunts
for a mismatch the fixed seeds of the suite do not hit.  usage: fuzz_extreme.py [n]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import test_gpu_fuzz as fz
from desman_amd import _lib
from desman_amd.synth import synth_counts
from oracle import cbind, ref_numpy as rn
import desman_amd.sampletau as st
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs0 = np.random.RandomState(12345)
bad = 0
pk = ["masked", "tiny", "subnormal", "zero_sample", "one_hot", "usual"]
ek = ["usual", "identity", "zero_row", "tiny", "subnormal"]
for i in range(n):
    V, S, G = int(rs0.randint(1, 300)), int(rs0.randint(1, 200)), int(rs0.randint(1, 17))
    a, b = pk[rs0.randint(len(pk))], ek[rs0.randint(len(ek))]
    rs = np.random.RandomState(i)
    counts, _, _ = synth_counts(V, S, min(G, 4), seed=i, depth_scale=float(rs.choice([0.05, 1.0, 30.0])))
    counts[rs.rand(V, S) < 0.1] = 0
    counts = np.ascontiguousarray(counts)
    tau0 = cbind.idx_to_onehot(rs.randint(4, size=(V, G)).astype(np.uint8))
    pi, eta = fz._extreme_pi(rs, S, G, a), fz._extreme_eta(rs, b)
    u = cbind.MT19937(i).uniform(2 * V * G)
    ref = tau0.copy()
    with np.errstate(all="ignore"):
        n_ref = [cbind.sample_tau_u(ref, pi, eta, counts, u[k * V * G:(k + 1) * V * G]) for k in range(2)]
    got = tau0.copy()
    st.initRNG(); st.setRNG(i)
    nn = [st.sample_tau(got, pi, eta, counts) for _ in range(2)]
    st.freeRNG()
    if nn != n_ref or not np.array_equal(got, ref):
        bad += 1
        print("MISMATCH tau shim", i, (V, S, G), a, b, nn, n_ref, int((got != ref).any(axis=(1, 2)).sum()), flush=True)
    # the same state through the context: log-probabilities and the log-likelihood of the state the sweep leaves (NaN / inf in the same places)
    with np.errstate(all="ignore"):
        r2 = tau0.copy()
        _, lp_ref = cbind.sample_tau_u(r2, pi, eta, counts, u[:V * G], want_logp=True)
        ll_ref = cbind.loglik(cbind.onehot_to_idx(r2), pi, eta, counts)
    ctx = _lib.Context(0)
    ctx.set_counts(counts); ctx.set_state(tau0, pi, eta); ctx.set_tau_rng(_lib.RNG_MT19937); ctx.set_mt_state(_lib.mt_seed_state(i))
    _, lp = ctx.sample_tau(want_logp=True)
    ll = ctx.loglik()[0]
    ctx.close()
    okl = np.array_equal(np.isnan(lp), np.isnan(lp_ref)) and np.array_equal(np.isinf(lp), np.isinf(lp_ref)) and \
        np.allclose(lp[np.isfinite(lp_ref)], lp_ref[np.isfinite(lp_ref)], rtol=1e-12, atol=0)
    okll = (np.isnan(ll) and np.isnan(ll_ref)) or ll == ll_ref or abs(ll - ll_ref) <= 1e-12 * abs(ll_ref)
    if not (okl and okll):
        bad += 1
        print("MISMATCH ctx logp/ll", i, (V, S, G), a, b, okl, ll, ll_ref, flush=True)
print("tau shim + context: %d cases, %d mismatches" % (n, bad))
bad = 0
for i in range(n):
    rs = np.random.RandomState(1000 + i)
    V = 40
    freq = rs.poisson(rs.choice([0.5, 5, 40, 4000]), size=(V, 4)).astype(np.int64) * int(rs.choice([1, 1, 1000, 10 ** 6]))
    freq[rs.rand(V) < 0.2] = 0
    m = rs.rand(V) < 0.2
    freq[m, 1:] = 0
    eta = [0.96 * np.eye(4) + 0.01, np.eye(4), np.full((4, 4), 0.25)][rs.randint(3)]
    maxA = np.argmax(freq, axis=1); ft = freq.copy(); ft[np.arange(V), maxA] = -1; maxB = np.argmax(ft, axis=1)
    ff = freq.astype(np.float64)
    p0 = np.minimum(freq.max(axis=1) / np.maximum(freq.sum(axis=1), 1), 0.99)
    opt = bool(rs.randint(2))
    with np.errstate(all="ignore"):
        g = _lib.lrt_step(ff, maxA, maxB, eta, 0.99, opt, p0)
        o = rn.lrt_step(ff, maxA, maxB, eta, 0.99, opt, p0)
    for name, x, y, tol in (("p", g[0], o[0], 1e-9), ("m", g[1], o[1], 1e-9), ("b", g[2], o[2], 1e-9)):
        fin = np.isfinite(y)
        ok = np.array_equal(np.isnan(x), np.isnan(y)) and np.array_equal(np.isfinite(x), fin) and np.allclose(x[fin], y[fin], rtol=1e-11, atol=tol) and np.array_equal(x[~fin & ~np.isnan(y)], y[~fin & ~np.isnan(y)])
        if not ok:
            bad += 1
            w = np.where(~(np.isclose(x, y, rtol=1e-11, atol=tol, equal_nan=True)))[0][:5]
            print("MISMATCH lrt", i, name, opt, w, x[w], y[w], freq[w].tolist(), flush=True)
print("lrt: %d cases, %d mismatches" % (n, bad))
