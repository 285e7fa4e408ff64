#!/bin/bash
# kernel-trace averages of the four Gibbs kernels over a scan of V (S, G fixed): looks for round quantisation.  usage: vscan_trace.sh S G V...
export DESMAN_HIP_LIB=${DESMAN_HIP_LIB:-$PWD/desman_amd/lib/libdesman_hip_ab.so}   # the experiment build: A/B switches compiled in (make -C desman_amd/csrc ab)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
S=$1; G=$2; shift 2
for V in "$@"; do
  rm -rf /tmp/vs_$V
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vs_$V -o t -- python scripts/prof_gibbs.py 200 $V $S $G > /dev/null 2>&1
  f=$(find /tmp/vs_$V -name 't_kernel_stats.csv' | head -1)
  python - "$f" $V <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
out = {}
for r in rows:
    n = r["Name"]
    for k in ("stats_agg_kernel", "stats_big_kernel", "dirichlet_kernel", "tau_kernel<"):
        if k in n and int(r["Calls"]) >= 100: out[k.strip("<")] = round(float(r["AverageNs"]) / 1000, 1)
print(sys.argv[2], out, "sum %.1f" % sum(out.values()), flush=True)
PY
done
