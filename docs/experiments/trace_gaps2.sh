#!/bin/bash
# kernel trace of the bench: where one Gibbs iteration's 107 us go -- kernel durations and the gaps between consecutive launches on the main stream
export DESMAN_HIP_LIB=${DESMAN_HIP_LIB:-$PWD/desman_amd/lib/libdesman_hip_ab.so}   # the experiment build: A/B switches compiled in (make -C desman_amd/csrc ab)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/tr; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --batch 0 > /dev/null 2>&1
f=$(find /tmp/tr -name 't_kernel_trace.csv' | head -1)
python - "$f" <<'PY'
import csv, sys, numpy as np
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "mt_fill" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = ["stats_agg", "stats_big", "dirichlet", "tau_kernel"]
seq = [(next((n for n in names if n in r["Kernel_Name"]), None), int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
seq = [x for x in seq if x[0]]
# the last 150 complete iterations
idx = [i for i, x in enumerate(seq) if x[0] == "stats_agg"][-151:]
dur = {n: [] for n in names}; gap = {n: [] for n in names}; it = []
for a, b in zip(idx[:-1], idx[1:]):
    blk = seq[a:b]
    if [x[0] for x in blk] != names: continue
    it.append((seq[b][1] - blk[0][1]) / 1000)
    for k, x in enumerate(blk):
        dur[x[0]].append((x[2] - x[1]) / 1000)
        nxt = blk[k + 1][1] if k + 1 < len(blk) else seq[b][1]
        gap[x[0]].append((nxt - x[2]) / 1000)
print("iterations", len(it), "mean %.2f us" % np.mean(it))
for n in names: print("%-11s duration %.2f   gap to the next launch %.2f" % (n, np.mean(dur[n]), np.mean(gap[n])))
print("sum of durations %.2f, sum of gaps %.2f" % (sum(np.mean(dur[n]) for n in names), sum(np.mean(gap[n]) for n in names)))
PY
