#!/bin/bash
# round 6: workgroups per list of the deferred-item launch (16 since round 3): what the almost always empty launch costs at config 3 by grid,
# in the 500-step form (settled chain) and in the driver's 20-step form (right after the NMF start: items ARE deferred there), at 50k x 96 x 12
# and at ten times the depth (experiment build: DESMAN_HIP_BIG_WGS)
export DESMAN_HIP_LIB=$PWD/desman_amd/lib/libdesman_hip_ab.so
{
for w in 16 8 4 2 1 16 4 1; do
echo -n "wgs $w  500-step: "; DESMAN_HIP_BIG_WGS=$w python bench.py --steps 500 --warmup 50 --no-cpu-baseline --batch 0 --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; print('%.2f us'%(d['ms_per_step']*1e3), {a: round(b,1) for a,b in k.items() if a in ('stats','stats_big','dirichlet','tau')})"
done
for w in 16 4 1 16 4 1; do
echo -n "wgs $w  20-step: "; DESMAN_HIP_BIG_WGS=$w python bench.py --steps 20 --warmup 5 --no-cpu-baseline --batch 0 --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; print('%.2f us'%(d['ms_per_step']*1e3), {a: round(b,1) for a,b in k.items() if a in ('stats','stats_big','dirichlet','tau')})"
done
for w in 16 4 1; do
echo -n "wgs $w  50k x 96 x 12: "; DESMAN_HIP_BIG_WGS=$w python bench.py --V 50000 --S 96 --G 12 --steps 100 --warmup 20 --no-cpu-baseline --batch 0 --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; print('%.2f us'%(d['ms_per_step']*1e3), {a: round(b,1) for a,b in k.items() if a in ('stats','stats_big','dirichlet','tau')})"
echo -n "wgs $w  config 3 at 10 x depth: "; DESMAN_HIP_BIG_WGS=$w python bench.py --depth-scale 10 --steps 100 --warmup 20 --no-cpu-baseline --batch 0 --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; print('%.2f us'%(d['ms_per_step']*1e3), {a: round(b,1) for a,b in k.items() if a in ('stats','stats_big','dirichlet','tau')})"
done
} 2>&1 | tee gpurun_out/r06_big_wgs.txt
