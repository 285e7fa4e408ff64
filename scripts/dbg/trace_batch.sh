#!/bin/bash
# rocprofv3 kernel trace of a batch of K config-3 chains (dsm_batch_gibbs_update): per-kernel averages and the gaps between launches.  usage: trace_batch.sh [K steps]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
K=${1:-4}; N=${2:-200}
cat > gpurun_out/_tb.py <<PY
import sys; sys.path.insert(0, '.')
from desman_amd import _lib
from desman_amd.synth import synth_counts, random_state
V, S, G = 10000, 64, 8
counts, _, _ = synth_counts(V, S, G, 1234)
cs = []
for k in range($K):
    c = _lib.Context(0); c.set_counts(counts); c.set_state(*random_state(V, S, G, seed=10 + k)); c.seed(100 + k, ctr_seed=77 + k); cs.append(c)
_lib.Context.batch_gibbs_update(cs, $N)
_lib.Context.batch_gibbs_update(cs, $N)
PY
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tb -o t -- python gpurun_out/_tb.py > gpurun_out/_tb.log 2>&1
python - <<'EOF2'
import csv, glob, statistics as st
f = (glob.glob("gpurun_out/tb/*/t_kernel_trace.csv") + glob.glob("gpurun_out/tb/t_kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]                       # the second call
main = [r for r in rows if not r["Kernel_Name"].startswith("mt_fill")]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
names = {}
for r in main: names.setdefault(r["Kernel_Name"].split("(")[0][:44], []).append(dur(r))
for k, v in sorted(names.items(), key=lambda x: -sum(x[1])): print("%-46s calls %5d  avg %7.1f us  total %9.1f us" % (k, len(v), sum(v) / len(v), sum(v)))
gaps = [(int(main[i + 1]["Start_Timestamp"]) - int(main[i]["End_Timestamp"])) / 1e3 for i in range(len(main) - 1)]
span = (int(main[-1]["End_Timestamp"]) - int(main[0]["Start_Timestamp"])) / 1e3
print("span %.1f us, kernels %.1f us, positive gaps %.1f us (median gap %.2f us), overlaps %.1f us" % (span, sum(dur(r) for r in main), sum(g for g in gaps if g > 0), st.median(gaps), -sum(g for g in gaps if g < 0)))
mt = [r for r in rows if r["Kernel_Name"].startswith("mt_fill")]
print("generator launches", len(mt), "avg %.1f us" % (sum(dur(r) for r in mt) / max(len(mt), 1)))
EOF2
rm -rf gpurun_out/tb gpurun_out/_tb.py gpurun_out/_tb.log
