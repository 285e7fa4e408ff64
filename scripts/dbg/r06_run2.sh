#!/bin/bash
mkdir -p gpurun_out
{
echo "== timeline (experiment build)"; python scripts/dbg/r06_s1_clocks.py 2>&1 | grep -E "kernel span|tables staged|four items|before epilogue|^end"
echo "== parity"; timeout 1500 python -m pytest tests -m gpu -x -q -k "stats or gibbs or batch or law or spec or shard" 2>&1 | grep -E "passed|failed|error" | tail -4
echo "== A/B"
bash scripts/dbg/lib_ab.sh "10000 64 8" "50000 96 12" "20000 32 5" "10000 96 8" "1000 16 5" -- r5=desman_amd/lib/libdesman_hip_r5.so new=desman_amd/lib/libdesman_hip.so
} > gpurun_out/r06_run2.txt 2>&1
cat gpurun_out/r06_run2.txt
