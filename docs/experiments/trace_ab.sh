#!/bin/bash
# rocprofv3 kernel-trace averages of the Gibbs loop's kernels for A/B variants (env VAR=VAL pairs given as arguments "name:ENV1=a,ENV2=b")
export DESMAN_HIP_LIB=${DESMAN_HIP_LIB:-$PWD/desman_amd/lib/libdesman_hip_ab.so}   # the experiment build: A/B switches compiled in (make -C desman_amd/csrc ab)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  O=/tmp/trace_$name; rm -rf $O
  env $(echo $envs | tr ',' ' ') rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python bench.py --steps 300 --warmup 50 --no-cpu-baseline --batch 0 > /tmp/bench_$name.json 2>/dev/null
  echo "== $name ($envs)  ms/step $(python -c "import json;print('%.4f'%json.loads(open('/tmp/bench_$name.json').read().strip().splitlines()[-1])['ms_per_step'])")"
  python - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(k in n for k in ("stats_agg", "stats_big", "dirichlet", "tau_kernel", "stage2")) and int(r["Calls"]) > 100:
        print("   %-60s calls %5s avg %.1f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
