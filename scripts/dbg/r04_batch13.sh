#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=30 2>&1 | tail -45 ) 2>&1 | tee $O/pytest_durations.txt
export DESMAN_HIP_LIB=$PWD/desman_amd/lib/libdesman_hip_ab.so
for cap in 64 128; do echo "== lean cap $cap"; DESMAN_HIP_LEAN_CAP=$cap python scripts/misfit_scan.py --gs 3,4,5,6 --out $O/tmp_cap.json 2>&1; done | tee $O/r04_lean_cap_pat.txt
