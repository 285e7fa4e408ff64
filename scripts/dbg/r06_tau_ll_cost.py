"""round 6: what the likelihood epilogue costs inside the sweep launch: tau_kernel<.,.,true,false> (sweep only: dsm_ctx_sample_tau) against
<true,true> (the Gibbs loop's) and <false,true> (likelihood only: dsm_ctx_loglik), per-kernel hipEvent times.  usage: r06_tau_ll_cost.py [V S G]"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts
from oracle import cbind
V, S, G = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (10000, 64, 8)
counts, tt, gg = synth_counts(V, S, G, 1234)
ctx = _lib.Context(0); ctx.set_counts(counts); ctx.seed(0)
ctx.set_state(cbind.idx_to_onehot(tt), np.ascontiguousarray(gg), 0.96 * np.eye(4) + 0.01)
ctx.gibbs_update(30)
def timed(fn, n):
    fn(); ctx.set_timing(True)
    for _ in range(n): fn()
    tm = ctx.get_timing(); ctx.set_timing(False)
    return {k: round(1e3 * ms / max(c, 1), 2) for k, (ms, c) in tm.items() if c}
print((V, S, G), "gibbs_update(100):", timed(lambda: ctx.gibbs_update(100), 3))
print((V, S, G), "sample_tau (sweep only):", timed(lambda: ctx.sample_tau(), 200))
print((V, S, G), "loglik (likelihood only):", timed(lambda: ctx.loglik(), 200))
