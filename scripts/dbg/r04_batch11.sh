#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) 2>&1 | tee $O/pytest_gpu_c.txt
for g in 12 10 8; do python scripts/dbg/chain_fp64.py $g 2>&1 | grep -E "share|burn-in|sampling|NMF"; done | tee $O/r04_chain_fp64.txt
python scripts/chain_phases.py --out $O/r04_chain_phases_c.json 2>&1 | grep "G=" | tee $O/r04_chain_phases_c.txt
python bench.py --V 10000 --S 8 --G 4 --steps 200 --warmup 50 --no-pmc --no-cpu-baseline --batch 0 --no-nmft | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('S=8', d['ms_per_step'], d['roofline']['kernels_us'], d['roofline'].get('tau_steps_fp64_frac'))"
python bench.py --depth-scale 0.05 --steps 200 --warmup 50 --no-pmc --no-cpu-baseline --batch 0 --no-nmft | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('depth x0.05', d['ms_per_step'], d['roofline']['kernels_us'], d['roofline'].get('tau_steps_fp64_frac'))"
