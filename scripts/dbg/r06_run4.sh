#!/bin/bash
mkdir -p gpurun_out
{
echo "== full GPU suite"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -4
echo "== A/B: ms per Gibbs iteration, per-kernel event times (r5 = round 5's library, new = round 6 with the sample-swizzled row map)"
bash scripts/dbg/lib_ab.sh "10000 64 8" "50000 96 12" "50000 96 8" "50000 96 4" "20000 32 5" "10000 96 8" "10000 192 8" "1000 16 5" -- r5=desman_amd/lib/libdesman_hip_r5.so new=desman_amd/lib/libdesman_hip.so
} > gpurun_out/r06_run4.txt 2>&1
cat gpurun_out/r06_run4.txt
