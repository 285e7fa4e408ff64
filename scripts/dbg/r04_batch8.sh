#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) 2>&1 | tee $O/pytest_gpu_b.txt
bash scripts/dbg/chain_trace.sh
python scripts/fit_chain_cost.py --out $O/r04_chain_cost_components_b.json 2>&1 | tee $O/r04_chain_cost_components_b.txt
