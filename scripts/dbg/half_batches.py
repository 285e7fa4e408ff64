"""a batch of four config-3 chains as ONE set of launches against TWO half-batches of two chains on their own streams
(two host threads, free-running, so that one half's latency-bound Dirichlet / deferred-item launches fall under the other half's stage 1
or sweep).  us per chain-iteration, same box, interleaved.   usage: half_batches.py [V S G steps]"""
import sys, time, threading; sys.path.insert(0, '.')
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts, random_state
a = [int(x) for x in sys.argv[1:]]
V, S, G, steps = (a + [10000, 64, 8, 300][len(a):])[:4]
counts, tau_true, gam_true = synth_counts(V, S, G, 1234)
def chain(k):
    c = _lib.Context(0); c.set_counts(counts); c.set_state(*random_state(V, S, G, seed=10 + k)); c.seed(100 + k, ctr_seed=77 + k); return c
cs = [chain(k) for k in range(4)]
_lib.Context.batch_gibbs_update(cs, steps)                                # warm: traces sized, tables placed
for rep in range(3):
    t0 = time.perf_counter(); _lib.Context.batch_gibbs_update(cs, steps); t4 = time.perf_counter() - t0
    def half(g): _lib.Context.batch_gibbs_update(g, steps)
    th = [threading.Thread(target=half, args=(cs[:2],)), threading.Thread(target=half, args=(cs[2:],))]
    t0 = time.perf_counter(); [t.start() for t in th]; [t.join() for t in th]; t22 = time.perf_counter() - t0
    t0 = time.perf_counter(); _lib.Context.batch_gibbs_update(cs[:2], steps); t2 = time.perf_counter() - t0
    print("batch of 4: %.1f us per chain-iteration | two half-batches of 2 on two streams: %.1f | one batch of 2 alone: %.1f" %
          (1e6 * t4 / (4 * steps), 1e6 * t22 / (4 * steps), 1e6 * t2 / (2 * steps)), flush=True)
