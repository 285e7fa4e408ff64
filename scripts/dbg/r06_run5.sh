#!/bin/bash
mkdir -p gpurun_out
{
echo "== compacted kernel block time line (experiment build) G=5 / G=3 on the six-strain 50k x 96 table"
python scripts/dbg/r06_big_clocks.py 2>&1 | tail -7; python scripts/dbg/r06_big_clocks.py 50000 96 3 6 2>&1 | tail -7
echo "== parity (stats / gibbs / batch / shard legs)"; timeout 1500 python -m pytest tests -m gpu -x -q -k "stats or gibbs or batch or law or spec or shard or words" 2>&1 | grep -E "passed|failed|error" | tail -4
echo "== bench per shape"
for shp in "10000 64 8 8" "50000 96 3 6" "50000 96 5 6" "50000 96 7 6" "50000 96 12 12" "10000 64 8 8 10"; do set -- $shp
  echo -n "$1 x $2 x $3 (from $4 strains, depth x${5:-1}): "; python bench.py --V $1 --S $2 --G $3 --true-G $4 --depth-scale ${5:-1} --steps 200 --warmup 100 --repeats 3 --no-cpu-baseline --batch 0 --no-pmc --no-nmft 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; print('%.2f us'%(d['ms_per_step']*1e3), {a: round(b,1) for a,b in k.items()})"
done
} 2>&1 | tee gpurun_out/r06_run5.txt
