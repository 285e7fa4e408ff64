#!/bin/bash
# kernel trace of the bench: per-launch durations of stats_agg_kernel / tau_kernel and whether an mt_fill launch overlaps them
export DESMAN_HIP_LIB=${DESMAN_HIP_LIB:-$PWD/desman_amd/lib/libdesman_hip_ab.so}   # the experiment build: A/B switches compiled in (make -C desman_amd/csrc ab)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/tr; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --batch 0 > /dev/null 2>&1
f=$(find /tmp/tr -name 't_kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
mt = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "mt_fill" in r["Kernel_Name"]]
for key in ("stats_agg_kernel", "tau_kernel<32, 2, true, true>", "dirichlet_kernel"):
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if key in r["Kernel_Name"]][-200:]
    d = np.array([(e - s) / 1000 for s, e in ks])
    ov = np.array([any(ms < e and me > s for ms, me in mt) for s, e in ks])
    ovs = np.array([any(ms <= s < me for ms, me in mt) for s, e in ks])      # the generator is running when the launch starts
    print(key, "n", len(d), "mean %.1f" % d.mean(), "min %.1f" % d.min(), "p10 %.1f p50 %.1f p90 %.1f" % tuple(np.percentile(d, [10, 50, 90])),
          "| overlapping mt: n %d mean %.1f | not: n %d mean %.1f | mt running at start: n %d mean %.1f" % (ov.sum(), d[ov].mean() if ov.any() else 0, (~ov).sum(), d[~ov].mean() if (~ov).any() else 0, ovs.sum(), d[ovs].mean() if ovs.any() else 0))
d = np.array([(e - s) / 1000 for s, e in mt]); print("mt_fill n", len(d), "mean %.1f" % d.mean(), "total %.0f us" % d.sum())
PY
