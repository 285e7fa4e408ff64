#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
( time python -m pytest tests/test_gpu_vshard.py tests/test_gpu_dist.py -x -q 2>&1 | tail -8 ) 2>&1 | tee $O/vshard_pytest.txt
for shp in "50000 96 12" "200000 64 8" "10000 64 8"; do python scripts/bench_vshard_comm.py $shp 100; done 2>&1 | tee $O/vshard_comm.txt
python scripts/chain_phases.py --out $O/chain_phases.json 2>&1 | grep -v "^$" | tee $O/chain_phases.txt
( time python bench.py --steps 20 --warmup 5 > $O/bench_default2.json ) 2>&1 | tail -3
python -c "
import json; d=json.load(open('$O/bench_default2.json')); print(d['ms_per_step'], d['ms_per_step_repeats']['all'], d['batch'])"
