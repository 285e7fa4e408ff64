#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) 2>&1 | tee $O/pytest_gpu_b.txt
python scripts/misfit_scan.py --gs 2,3,4,5,6 --out $O/r04_misfit_scan_spec4.json 2>&1 | tee $O/r04_misfit_scan_spec4b.txt
python scripts/chain_phases.py --out $O/r04_chain_phases_b.json 2>&1 | grep -v "^$" | tee $O/r04_chain_phases_b.txt
