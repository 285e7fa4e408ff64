#!/bin/bash
# PMC passes of the NMFT update kernel at a large shape.  usage: pmc_nmft_big.sh V S G  -> gpurun_out/pmc_nmft_big.csv
export DESMAN_HIP_LIB=${DESMAN_HIP_LIB:-$PWD/desman_amd/lib/libdesman_hip_ab.so}   # the experiment build: A/B switches compiled in (make -C desman_amd/csrc ab)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_nb; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o p -- python scripts/prof_nmft.py $1 $2 $3 40 > $O/l1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $O/p2 -o p -- python scripts/prof_nmft.py $1 $2 $3 40 > $O/l2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/p3 -o p -- python scripts/prof_nmft.py $1 $2 $3 40 > $O/l3.log 2>&1
python scripts/summarize_pmc.py gpurun_out/pmc_nmft_big.csv $O/p1 $O/p2 $O/p3
python - <<'PY'
import csv
for r in csv.DictReader(open("gpurun_out/pmc_nmft_big.csv")):
    if "mfma" in r["kernel"] or "reduce" in r["kernel"]:
        print(r["kernel"][:40], {k: v for k, v in r.items() if k != "kernel" and v})
PY
tail -2 $O/l1.log
