"""one case of scripts/dbg/fuzz_extreme.py's tau-shim leg in detail: the first (v, g) where the device sweep leaves the oracle, with both sides' log-probabilities.  usage: fuzz_extreme_case.py i V S G pi_kind eta_kind"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import test_gpu_fuzz as fz
from desman_amd import _lib
from desman_amd.synth import synth_counts
from oracle import cbind
i, V, S, G = (int(x) for x in sys.argv[1:5]); a, b = sys.argv[5:7]
rs0 = np.random.RandomState(12345)
# replay the generator of fuzz_extreme.py up to case i to get the depth scale
rs = np.random.RandomState(i)
counts, _, _ = synth_counts(V, S, min(G, 4), seed=i, depth_scale=float(rs.choice([0.05, 1.0, 30.0])))
counts[rs.rand(V, S) < 0.1] = 0
counts = np.ascontiguousarray(counts)
tau0 = cbind.idx_to_onehot(rs.randint(4, size=(V, G)).astype(np.uint8))
pi, eta = fz._extreme_pi(rs, S, G, a), fz._extreme_eta(rs, b)
u = cbind.MT19937(i).uniform(V * G)
ref = tau0.copy()
with np.errstate(all="ignore"):
    n_ref, lp_ref = cbind.sample_tau_u(ref, pi, eta, counts, u, want_logp=True)
for screen in (True, False):
    ctx = _lib.Context(0)
    ctx.set_counts(counts); ctx.set_state(tau0, pi, eta)
    ctx.set_tau_rng(_lib.RNG_MT19937); ctx.set_mt_state(_lib.mt_seed_state(i))
    ctx.set_tau_screen(screen)
    n, lp = ctx.sample_tau(want_logp=True)
    got = ctx.get_state()[0]
    ctx.close()
    bad = np.argwhere((got != ref).any(axis=2))
    print("screen", screen, "nchange", n, n_ref, "mismatching (v,g):", len(bad))
    np.set_printoptions(precision=17, linewidth=200)
    for v, g in bad[:4]:
        print(" v", v, "g", g, "u", u[v * G + g], "tau0", cbind.onehot_to_idx(tau0)[v], "ref", cbind.onehot_to_idx(ref)[v, g], "got", cbind.onehot_to_idx(got)[v, g])
        print("   oracle logp", lp_ref[v, g]); print("   device logp", lp[v, g])
    # the first (v, g) whose log-probabilities differ in NaN / inf pattern or by more than 1e-9 relative
    with np.errstate(all="ignore"):
        d = ~(np.isclose(lp, lp_ref, rtol=1e-9, atol=0, equal_nan=True))
    w = np.argwhere(d.any(axis=2))
    print(" (v,g) with different log-probabilities:", len(w))
    for v, g in w[:4]:
        print("  v", v, "g", g, "oracle", lp_ref[v, g], "device", lp[v, g])
print("pi", pi[:3]); print("eta", eta)
