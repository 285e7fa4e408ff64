#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_batch.py tests/test_gpu_edges.py tests/test_gpu_vshard.py tests/test_gpu_host.py -x -q 2>&1 | tail -15 ) 2>&1 | tee $O/neartie_pytest.txt
python scripts/dbg/chain_fp64.py 12 2>&1 | grep -E "share|burn-in|sampling|NMF" | tee $O/r04_chain_fp64.txt
python scripts/dbg/chain_fp64.py 8 2>&1 | grep -E "share|burn-in|sampling|NMF" | tee -a $O/r04_chain_fp64.txt
python bench.py --steps 500 --warmup 50 --no-pmc --no-cpu-baseline --batch 0 > $O/bench_nt_cfg3.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_nt_cfg3.json')); print('cfg3', d['ms_per_step'], d['roofline']['kernels_us'], d['roofline'].get('tau_steps_fp64_frac'))"
python bench.py --V 50000 --S 96 --G 12 --steps 100 --warmup 20 --no-pmc --no-cpu-baseline --batch 0 > $O/bench_nt_cfg5.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_nt_cfg5.json')); print('cfg5', d['ms_per_step'], d['roofline']['kernels_us'], d['roofline'].get('tau_steps_fp64_frac'))"
python scripts/chain_phases.py --out $O/r04_chain_phases_c.json 2>&1 | grep "G=" | tee $O/r04_chain_phases_c.txt
