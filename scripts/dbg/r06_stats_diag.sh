#!/bin/bash
mkdir -p gpurun_out
{
python scripts/dbg/r06_stats_diag.py desman_amd/lib/libdesman_hip_r5ab.so
python scripts/dbg/r06_stats_diag.py desman_amd/lib/libdesman_hip_ab.so
} > gpurun_out/r06_stats_diag.txt 2>&1
# PMC: r5 vs new (product libs), the Gibbs loop at config 3
cd /tmp && export TMPDIR=/tmp
for lib in r5 new; do
  L=$GRAFT_REPO_ROOT/desman_amd/lib/libdesman_hip.so; [ $lib = r5 ] && L=$GRAFT_REPO_ROOT/desman_amd/lib/libdesman_hip_r5.so
  for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SMEM SQ_IFETCH"; do
    d=/tmp/pmc_${lib}_$(echo $grp | cut -d' ' -f1)
    DESMAN_HIP_LIB=$L timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -o p -- python $GRAFT_REPO_ROOT/scripts/prof_gibbs.py 30 > /dev/null 2>&1
    python - "$d" "$lib" <<'PY' >> $GRAFT_REPO_ROOT/gpurun_out/r06_stats_diag.txt
import csv, glob, sys, collections
d, lib = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        if "stats_agg" in k or "tau_kernel" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print("pmc", lib, k, {n: round(sum(v) / len(v)) for n, v in c.items()}, "n", max(len(v) for v in c.values()))
PY
  done
done
cd $GRAFT_REPO_ROOT; tail -80 gpurun_out/r06_stats_diag.txt
