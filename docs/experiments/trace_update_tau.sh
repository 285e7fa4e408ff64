#!/bin/bash
# rocprofv3 kernel trace of updateTau at V S G (default 50000 96 8): duration of the sweep launches in situ and the gaps between them
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
V=${1:-50000}; S=${2:-96}; G=${3:-8}
sed "s/V, S, G, n = 10000, 64, 8, 200/V, S, G, n = $V, $S, $G, 100/" scripts/prof_update_tau.py > gpurun_out/_ut.py
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ut -o t -- python gpurun_out/_ut.py > gpurun_out/_ut.log 2>&1
tail -1 gpurun_out/_ut.log
python - <<'EOF'
import csv, glob, statistics as st
f = (glob.glob("gpurun_out/ut/*/t_kernel_trace.csv") + glob.glob("gpurun_out/ut/t_kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
tau = [r for r in rows if r["Kernel_Name"].startswith("void tau_kernel<") and "true, true" in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in tau]
g = [(int(tau[i + 1]["Start_Timestamp"]) - int(tau[i]["End_Timestamp"])) / 1e3 for i in range(len(tau) - 1)]
print(len(tau), "sweep launches: duration median %.1f us (last 100: %.1f), gap to the next median %.1f us (last 99: %.1f, max %.1f)"
      % (st.median(d), st.median(d[-100:]), st.median(g), st.median(g[-99:]), max(g)))
mt = [r for r in rows if r["Kernel_Name"].startswith("mt_fill")]
print(len(mt), "generator launches, median %.1f us" % st.median([(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in mt]))
names = {}
for r in rows[-400:]:
    names[r["Kernel_Name"][:50]] = names.get(r["Kernel_Name"][:50], 0) + 1
print(names)
EOF
rm -rf gpurun_out/ut gpurun_out/_ut.py gpurun_out/_ut.log
