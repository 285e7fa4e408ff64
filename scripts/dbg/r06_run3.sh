#!/bin/bash
mkdir -p gpurun_out
{
echo "== full GPU suite"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -4
echo "== timeline (experiment build, config 3, generating state, table at +256 B)"; python scripts/dbg/r06_s1_clocks.py 2>&1
echo "== A/B: ms per Gibbs iteration, per-kernel event times"
bash scripts/dbg/lib_ab.sh "10000 64 8" "50000 96 12" "50000 96 8" "20000 32 5" "10000 96 8" -- r5=desman_amd/lib/libdesman_hip_r5.so new=desman_amd/lib/libdesman_hip.so
} > gpurun_out/r06_run3.txt 2>&1
cat gpurun_out/r06_run3.txt
