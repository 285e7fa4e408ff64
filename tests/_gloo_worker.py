"""worker for tests/test_chains_gloo.py: world_size-2 run of the chain scheduler on CPU."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist  # noqa: E402

from desman_amd import chains  # noqa: E402


def fake_run(spec):
    # deterministic stand-in for a GPU chain: the scheduler/gather is what is under test
    g, s = spec["G"], spec["seed"]
    return dict(G=g, seed=s, G_final=g - (s % 2), lp_star=-1000.0 * g - s, mean_dev=2000.0 * g + s, iters=10,
                wall_s=0.001 * g)


_calls = {}


def flaky_run(spec):
    """chain (G=3, seed=1) raises on every attempt, chain (G=5, seed=2) only on its first: the scheduler must
    re-queue once, mark the first as failed, and still reach the gather on both ranks"""
    key = (spec["G"], spec["seed"])
    _calls[key] = _calls.get(key, 0) + 1
    if key == (3, 1) or (key == (5, 2) and _calls[key] == 1):
        raise RuntimeError("injected failure %s attempt %d" % (key, _calls[key]))
    return fake_run(spec)


if __name__ == "__main__":
    dist.init_process_group("gloo")
    specs = chains.sweep_specs(range(2, 7), 3, V=1000, S=16)
    if len(sys.argv) > 2 and sys.argv[2] == "cfg5":
        # BASELINE config 5 on a full node: 55 chains (g = 2..12 x 5 seeds) at V = 50k, S = 96 over 8 ranks, unbatched and
        # with the replicates of a G value as batched units; every rank records which chains it was given
        specs = chains.sweep_specs(range(2, 13), 5, V=50000, S=96)
        ran = []

        def rec_run(spec):
            ran.append(spec["chain"])
            return fake_run(spec)

        def rec_batch(group):
            ran.extend(sp["chain"] for sp in group)
            return [fake_run(sp) for sp in group]
        recs = chains.run_chains(specs, rec_run, dist)
        one = list(ran)
        del ran[:]
        recs_b = chains.run_chains(specs, rec_run, dist, batch_fn=rec_batch, batch=5)
        with open(os.path.join(sys.argv[1], "cfg5_%d.json" % dist.get_rank()), "w") as f:
            json.dump(dict(recs=recs, recs_b=recs_b, one=one, batched=list(ran)), f)
        dist.destroy_process_group()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "queue":
        # the work-queue schedule: rank 1 is "slow" (every chain sleeps ten times longer there), the counter lives in the rendezvous
        # store; one chain fails once (re-queued on the rank that drew it).  Every chain must run exactly once over both ranks.
        import time
        from torch.distributed.distributed_c10d import _get_default_store
        ran = []

        def timed_run(spec):
            ran.append(spec["chain"])
            time.sleep(0.004 * (10 if dist.get_rank() == 1 else 1))
            return flaky_run(spec) if (spec["G"], spec["seed"]) == (5, 2) else fake_run(spec)
        q = chains.WorkQueue(store=_get_default_store(), key="q_test")
        recs = chains.run_chains(specs, timed_run, dist, queue=q, concurrency=2)
        with open(os.path.join(sys.argv[1], "queue%d.json" % dist.get_rank()), "w") as f:
            json.dump(dict(recs=recs, ran=ran), f)
        dist.destroy_process_group()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "flaky":
        recs = chains.run_chains(specs, flaky_run, dist)
        with open(os.path.join(sys.argv[1], "flaky%d.json" % dist.get_rank()), "w") as f:
            json.dump(dict(recs=recs, calls={"%d_%d" % k: v for k, v in _calls.items()}), f)
        dist.destroy_process_group()
        sys.exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "batch":
        # replicate chains scheduled and run as units of up to 2; the unit of G = 4 fails as a batch and falls back to its
        # chains one by one
        units_run = []

        def fake_batch(group):
            units_run.append([(sp["G"], sp["seed"]) for sp in group])
            if group[0]["G"] == 4:
                raise RuntimeError("injected batch failure")
            return [fake_run(sp) for sp in group]
        recs = chains.run_chains(specs, fake_run, dist, batch_fn=fake_batch, batch=2)
        with open(os.path.join(sys.argv[1], "batch%d.json" % dist.get_rank()), "w") as f:
            json.dump(dict(recs=recs, units=units_run), f)
        dist.destroy_process_group()
        sys.exit(0)
    recs = chains.run_chains(specs, fake_run, dist)
    bins = chains.lpt_assign([s["cost"] for s in specs], dist.get_world_size())
    out = dict(rank=dist.get_rank(), recs=recs, mine=bins[dist.get_rank()])
    with open(os.path.join(sys.argv[1], "rank%d.json" % dist.get_rank()), "w") as f:
        json.dump(out, f)
    dist.destroy_process_group()
