"""Times the mu/E pass (stage 1 / stage 2 / v1) on a random state and on the generating state.
usage: python scripts/prof_stats.py [V S G reps depth_scale states versions]"""
import sys; sys.path.insert(0, '.')
import numpy as np
from desman_amd import _lib
from desman_amd.synth import synth_counts, random_state
from oracle import cbind
a = sys.argv[1:]
V, S, G = (int(a[0]), int(a[1]), int(a[2])) if len(a) >= 3 else (10000, 64, 8)
reps = int(a[3]) if len(a) > 3 else 20
scale = float(a[4]) if len(a) > 4 else 1.0
states = a[5].split(",") if len(a) > 5 else ["truth", "random"]
vers = a[6].split(",") if len(a) > 6 else ["v2", "v1"]
counts, tt, gg = synth_counts(V, S, G, 1234, depth_scale=scale)
ctx = _lib.Context(0); ctx.set_counts(counts); ctx.seed(1)
for name in states:
    if name == "random":
        tau, gamma, eta = random_state(V, S, G, seed=1)
    else:
        tau, gamma, eta = cbind.idx_to_onehot(tt), np.ascontiguousarray(gg), 0.96 * np.eye(4) + 0.01
    ctx.set_state(tau, gamma, eta)
    for force in [v == "v1" for v in vers]:
        ctx.force_stats_spec(1 if force else 2)
        ctx.sample_stats(0)
        ctx.set_timing(True)
        for it in range(reps):
            ctx.sample_stats(it + 1)
        tm = ctx.get_timing(); ctx.set_timing(False)
        print(name, "v1" if force else "v2", {k: round(1e3 * ms / max(n, 1), 1) for k, (ms, n) in tm.items() if n})
ctx.force_stats_spec(0)
