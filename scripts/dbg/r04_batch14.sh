#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_batch.py tests/test_gpu_host.py tests/test_gpu_edges.py -x -q 2>&1 | tail -12 ) 2>&1 | tee $O/nt_pytest.txt
for g in 12 10 8; do python scripts/dbg/chain_fp64.py $g 2>&1 | grep -E "share|burn-in|sampling|NMF"; done | tee $O/r04_chain_fp64_nt.txt
python scripts/chain_phases.py --out $O/r04_chain_phases_nt.json 2>&1 | grep "G=" | cut -c1-170 | tee $O/r04_chain_phases_nt.txt
python bench.py --steps 500 --warmup 50 --no-pmc --no-cpu-baseline --batch 0 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cfg3', d['ms_per_step'], d['roofline']['kernels_us'])"
