#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_batch.py tests/test_gpu_host.py -x -q 2>&1 | tail -15 ) 2>&1 | tee $O/flat_pytest.txt
python scripts/misfit_scan.py --gs 7,8,10,12 --out $O/r04_misfit_scan_flat.json 2>&1 | tee $O/r04_misfit_scan_flat.txt
python scripts/misfit_scan.py --V 10000 --S 64 --true-G 4 --gs 6,8 --out $O/r04_misfit_scan_flat_10k.json 2>&1 | tee -a $O/r04_misfit_scan_flat.txt
python bench.py --steps 500 --warmup 50 --no-pmc --no-cpu-baseline --batch 0 > $O/bench_flat_cfg3.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_flat_cfg3.json')); print('cfg3', d['ms_per_step'], d['roofline']['kernels_us'])"
python bench.py --V 50000 --S 96 --G 12 --steps 100 --warmup 20 --no-pmc --no-cpu-baseline --batch 0 > $O/bench_flat_cfg5.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench_flat_cfg5.json')); print('cfg5', d['ms_per_step'], d['roofline']['kernels_us'])"
