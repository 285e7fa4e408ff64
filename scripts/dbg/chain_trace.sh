#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
for G in 12 8; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ctrace$G -o t -- python scripts/dbg/chain_kernels.py $G > $O/chain_trace_G$G.log 2>&1
  f=$(ls $O/ctrace$G/*/t_kernel_stats.csv $O/ctrace$G/t_kernel_stats.csv 2>/dev/null | head -1)
  echo "== G=$G"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]: print("%-70s calls %6s avg %9.1f us total %8.1f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
  rm -rf $O/ctrace$G
done 2>&1 | tee $O/r04_chain_trace.txt
