"""one NMFT case of the fuzzer (scripts/dbg/fuzz_nmft.py) in detail: where the device's factors leave the oracle's.  usage: nmft_case.py V S G fix [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from desman_amd import _lib
if os.environ.get('NM_OLD_LIB'): _lib.SIGNATURES.pop('dsm_debug_ntab_probes', None)      # (a library of an earlier round)
from desman_amd.synth import synth_counts
from oracle import cbind, ref_numpy as rn
V, S, G, fix = [int(x) for x in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 12
counts, _, _ = synth_counts(V, S, min(G, 4), seed=V + S)
tau0, gam0 = rn.nmft_random_initialize(np.random.RandomState(V + 3 * S + G), V, S, G)
F = cbind.nmft_freq(counts)
for k in range(1, iters + 1):
    tc, gc = tau0.copy(), gam0.copy()
    n_ref, tr_ref = (cbind.nmft_factorize_tau if fix else cbind.nmft_factorize)(F, tc, gc, max_iter=k, min_change=0.0)
    c = _lib.Context(0); c.set_counts(counts); c.nmft_set(tau0, gam0)
    n, tr = c.nmft_factorize(max_iter=k, min_change=0.0, fix_gamma=bool(fix))
    t, g = c.nmft_get(); c.close()
    bad = np.argwhere(~np.isfinite(t) | (np.abs(t - tc) > 1e-6 * np.abs(tc) + 1e-12))
    print("updates %d: device n %d (oracle %d), trace tail %s vs %s, bad tau entries %d" % (k, n, n_ref, tr[-2:], tr_ref[-2:], len(bad)), flush=True)
    if len(bad):
        for r, gg in bad[:12]:
            a, v = divmod(int(r), V)
            print("   row %d = (base %d, variant %d [quad %d, vv %d]) g %d: device %r oracle %r   start %r" % (r, a, v, v // 4, v % 4, gg, t[r, gg], tc[r, gg], tau0[r, gg]))
        vs = sorted({int(r) % V for r, _ in bad})
        print("   variants:", vs[:40], "... of", len(vs))
        break
