"""Seeded subsets of the fuzzers (scripts/dbg/fuzz_gibbs.py, fuzz_nmft.py) and the cross-process determinism soak
(scripts/dbg/soak_determinism.py) inside the driver-run suite: the boundaries of every kernel selection first, then random shapes.

* Gibbs: the WHOLE device iteration against the oracle composition in the reference's order (HaploSNP_Sampler.py:341-358; tau sweep
  c_sample_tau.c:95-204 bit for bit) -- lane-group widths S = 16/17, 32/33, 48, 64/65, 96/97, 128/129, 192/193, 256/257, 384/385, the
  stage-2 switch G = 9/10, the aggregated specs' end G = 16/17, the single-haplotype and single-sample cases.
* NMFT: factors, update count, objective trace and get_tau against the C oracle (Init_NMFT.py:98-245) at the tile boundaries of the
  matrix-core kernels (S = 16 k, 128/129, 256/257, 512; G = 4/5, 8/9, 12/13, 16; V not a multiple of four).
* Determinism: three fresh processes -- two with the run-time placement / ordering heuristics on, one with them off -- end on the
  same bits: where and when things run differs from process to process, never what is computed."""
import os
import subprocess
import sys

import numpy as np
import pytest

from desman_amd import _lib
from desman_amd.synth import synth_counts
from oracle import cbind, ref_numpy as rn

import test_gpu_parity as tp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gibbs_shapes():
    rs = np.random.RandomState(11)
    edge = [(150, S, 4) for S in (1, 16, 17, 33, 48, 65, 97, 129, 193, 257, 385)] + [(260, 24, G) for G in (1, 9, 10, 16, 17)] + \
           [(V, 64, 8) for V in (1, 3, 257)]
    rnd = [(int(rs.randint(1, 700)), int(rs.randint(1, 140)), int(rs.randint(1, 13))) for _ in range(6)]
    return edge + rnd


@pytest.mark.parametrize("V,S,G", _gibbs_shapes())
def test_fuzz_whole_gibbs_iteration_against_the_oracle(V, S, G):
    for spec in (2, 1) if (V + S + G) % 3 else (3, 1):          # every third shape takes the table exp / log variant of the aggregated spec
        if G > 16 and spec >= 2:
            continue                                            # the aggregated specifications stop at 16 haplotypes
        ctx = _lib.Context(0)
        try:
            tp.test_gibbs_update_is_self_consistent_with_oracle(ctx, V, S, G, 3, spec)
        finally:
            ctx.close()


def _nmft_shapes():
    rs = np.random.RandomState(7)
    edge = [(203, S, 3) for S in (1, 16, 17, 65, 97, 128, 129, 257, 512)] + [(77, S, 13) for S in (15, 64, 112, 193, 289)] + \
           [(501, 200, G) for G in (1, 5, 9, 16)] + [(V, 300, 7) for V in (1, 3, 5, 1025)]
    rnd = [(int(rs.randint(1, 3000)), int(rs.randint(1, 513)), int(rs.randint(1, 17))) for _ in range(6)]
    return edge + rnd


@pytest.mark.parametrize("V,S,G", _nmft_shapes())
def test_fuzz_nmft_against_the_oracle(V, S, G):
    counts, _, _ = synth_counts(V, S, min(G, 4), seed=V + S)
    tau0, gam0 = rn.nmft_random_initialize(np.random.RandomState(V + 3 * S + G), V, S, G)
    F = cbind.nmft_freq(counts)
    for fix_gamma in (False, True):
        tc, gc = tau0.copy(), gam0.copy()
        n_ref, tr_ref = (cbind.nmft_factorize_tau if fix_gamma else cbind.nmft_factorize)(F, tc, gc, max_iter=12, min_change=1e-5)
        if np.isnan(tc).any() or np.isnan(tr_ref[:n_ref]).any():
            continue          # a start with exact zeros in all four bases of a haplotype: 0/0 in the reference as well
        c = _lib.Context(0)
        c.set_counts(counts)
        c.nmft_set(tau0, gam0)
        n, tr = c.nmft_factorize(max_iter=12, min_change=1e-5, fix_gamma=fix_gamma)
        t, g = c.nmft_get()
        oh = c.nmft_get_tau()
        c.close()
        assert n == n_ref
        np.testing.assert_allclose(tr[:n], tr_ref[:n_ref], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(t, tc, rtol=1e-6, atol=1e-12)
        np.testing.assert_allclose(g, gc, rtol=1e-6, atol=1e-12)
        assert np.array_equal(oh, cbind.idx_to_onehot(cbind.nmft_get_tau(np.ascontiguousarray(t), G)))


_SOAK = r'''
import sys, hashlib; sys.path.insert(0, %r)
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
V, S, G = (int(x) for x in sys.argv[1:4])
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
for n in (7, 300, 1, 193, 20, 479):
    ctx.gibbs_update(n)
    tr = ctx.get_trace()
    for k in ("ll", "lp", "nchange", "gamma", "eta"): h.update(np.ascontiguousarray(tr[k]).tobytes())
    tt, gg, ee = ctx.get_state(); h.update(tt.tobytes()); h.update(gg.tobytes()); h.update(ee.tobytes())
print(h.hexdigest(), tr["ll"][-1])
''' % ROOT


@pytest.mark.parametrize("V,S,G", [(10000, 64, 8), (6000, 96, 5)])
def test_three_processes_end_on_the_same_bits(V, S, G):
    """1000 iterations in calls of uneven length after a 300-update NMF start (the benchmark's chain; the second shape takes the
    register-lean sweep and a table whose rows are not whole 256 B blocks).  DESMAN_HIP_NTAB_TUNE=0 / DESMAN_HIP_TAU_ORDER=0 switch the
    measured table place and the fp64-blocks-first order off: the bits may not depend on either."""
    outs = []
    for env in ({}, {"DESMAN_HIP_NTAB_TUNE": "0", "DESMAN_HIP_TAU_ORDER": "0"}, {}):
        r = subprocess.run([sys.executable, "-c", _SOAK, str(V), str(S), str(G)], env=dict(os.environ, **env), capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.split()[0])
    assert len(set(outs)) == 1 and len(outs[0]) == 64, outs
