#!/bin/bash
# round 6: the sweep launch with fewer workgroups than variant groups (persistent-style: resident workgroups stride over the variants)
export DESMAN_HIP_LIB=$PWD/desman_amd/lib/libdesman_hip_ab.so
for shp in "10000 64 8" "50000 96 12" "20000 32 5"; do set -- $shp
for g in 0 1536 1024 768 704 640 512 384; do echo -n "$shp grid $g: "; DESMAN_HIP_TAU_GRID=$g python bench.py --V $1 --S $2 --G $3 --steps 200 --warmup 30 --no-cpu-baseline --batch 0 --no-pmc --no-nmft 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; print('%.2f us'%(d['ms_per_step']*1e3), 'tau %.1f'%k['tau'], d['roofline']['tau_launch'])"; done; done 2>&1 | tee gpurun_out/r06_tau_grid.txt
