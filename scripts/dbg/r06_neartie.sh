#!/bin/bash
# round 6: an over-fitted chain (G = 12 / 10 on a six-strain 50k x 96 table): does a rare haplotype's step that the difference screen
# left open still get settled by the totals screen?  DESMAN_HIP_NT_SKIP_TOTALS=1 sends it straight to fp64 (experiment build).
export DESMAN_HIP_LIB=$PWD/desman_amd/lib/libdesman_hip_ab.so
for G in 12 10 8; do for sk in 0 1; do
  echo -n "G $G true 6 skip_totals $sk: "
  if [ $sk = 1 ]; then export DESMAN_HIP_NT_SKIP_TOTALS=1; else unset DESMAN_HIP_NT_SKIP_TOTALS; fi
  python bench.py --V 50000 --S 96 --G $G --true-G 6 --steps 100 --warmup 400 --repeats 3 --no-cpu-baseline --batch 0 --no-pmc --no-nmft 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels_us']; print('%.1f us/it'%(d['ms_per_step']*1e3), 'tau %.1f'%k['tau'], 'fp64 frac %.4f'%r['tau_steps_fp64_frac'])"
done; done 2>&1 | tee gpurun_out/r06_neartie.txt
