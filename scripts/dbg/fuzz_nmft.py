"""NMFT device path against the C oracle over random shapes and the boundaries of the kernel selection (S = 16 k, 128/129, 192/193, 256/257,
288/289, 384/385, 512; G = 1, 4/5, 8/9, 12/13, 16; V not a multiple of four): factors, update count, objective trace, get_tau.
usage: fuzz_nmft.py [n_random] [seed] [big]      big: V = 13 000 .. 60 000, S <= 160 -- a wavefront of the update kernel then walks several
quads (the look-ahead loop, the clamped addressing of its last round), four updates per case;
extreme: start values across the exponent range (tau entries x 1e-150 .. 5e-324 and exact zeros, abundances x 1e-120 .. 1e-300 at random places)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
from oracle import cbind, ref_numpy as rn
n_rand = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
shapes = [(V, S, G) for S in (1, 15, 16, 17, 64, 65, 96, 97, 112, 128, 129, 192, 193, 256, 257, 288, 289, 384, 385, 511, 512) for V, G in ((203, 3), (77, 13))]
shapes += [(501, 200, G) for G in (1, 2, 4, 5, 8, 9, 12, 13, 16)] + [(V, 300, 7) for V in (1, 2, 3, 4, 5, 63, 1025)]
shapes += [(int(rs.randint(1, 3000)), int(rs.randint(1, 513)), int(rs.randint(1, 17))) for _ in range(n_rand)]
BIG = len(sys.argv) > 3 and sys.argv[3] == "big"
if BIG:
    shapes = [(int(rs.randint(13000, 60000)), int(rs.choice([rs.randint(1, 161), rs.choice([16, 32, 48, 64, 65, 80, 81, 96, 97, 112, 128])])), int(rs.randint(1, 17))) for _ in range(n_rand)]
MAXIT = 4 if BIG else 12
bad = skipped = 0
for V, S, G in shapes:
    counts, _, _ = synth_counts(V, S, min(G, 4), seed=V + S)
    tau0, gam0 = rn.nmft_random_initialize(np.random.RandomState(V + 3 * S + G), V, S, G)
    if len(sys.argv) > 3 and sys.argv[3] == "extreme":
        for a, lo in ((tau0, -323.0), (gam0, -300.0)):
            m = rs.rand(*a.shape) < rs.choice([0.002, 0.02, 0.2])
            a[m] *= 10.0 ** rs.uniform(lo, -100.0, size=int(m.sum()))
            a[rs.rand(*a.shape) < 0.001] = 0.0
    F = cbind.nmft_freq(counts)
    for fix_gamma in (False, True):
        tc, gc = tau0.copy(), gam0.copy()
        n_ref, tr_ref = (cbind.nmft_factorize_tau if fix_gamma else cbind.nmft_factorize)(F, tc, gc, max_iter=MAXIT, min_change=1e-5)
        if np.isnan(tc).any() or np.isnan(tr_ref[:n_ref]).any(): skipped += 1; continue      # a start with exact zeros in all four bases of a haplotype: 0/0 in the reference as well
        c = _lib.Context(0); c.set_counts(counts); c.nmft_set(tau0, gam0)
        n, tr = c.nmft_factorize(max_iter=MAXIT, min_change=1e-5, fix_gamma=fix_gamma)
        t, g = c.nmft_get(); oh = c.nmft_get_tau(); c.close()
        ok = n == n_ref and np.allclose(tr[:n], tr_ref[:n_ref], rtol=1e-9, atol=1e-9) and np.allclose(t, tc, rtol=1e-6, atol=1e-12) and np.allclose(g, gc, rtol=1e-6, atol=1e-12) \
            and np.array_equal(oh, cbind.idx_to_onehot(cbind.nmft_get_tau(np.ascontiguousarray(t), G)))
        if not ok:
            bad += 1
            print("MISMATCH V=%d S=%d G=%d fix_gamma=%s: n %d vs %d, trace diff %.3g, tau diff %.3g, gamma diff %.3g" % (V, S, G, fix_gamma, n, n_ref,
                  np.max(np.abs(tr[:min(n, n_ref)] - tr_ref[:min(n, n_ref)]) / np.abs(tr_ref[:min(n, n_ref)])), np.max(np.abs(t - tc)), np.max(np.abs(g - gc))), flush=True)
print("shapes %d (x 2: gamma updating / fixed), cases skipped (NaN in the oracle too) %d, mismatches %d" % (len(shapes), skipped, bad))
