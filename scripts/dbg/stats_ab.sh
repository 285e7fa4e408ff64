#!/bin/bash
run() { python bench.py --steps 300 --warmup 50 --no-cpu-baseline --batch 0 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' ms/step %.4f'%d['ms_per_step'], {k:round(v,1) for k,v in d['roofline']['kernels_us'].items()}, d['roofline']['stats_spec'])"; }
for xcd in 0 1; do for spec in 2 3; do
  echo "xcd=$xcd spec=$spec regg=0"
  DESMAN_HIP_NTAB_XCD=$xcd DESMAN_HIP_STATS_SPEC=$spec DESMAN_HIP_STATS_REGG=0 run
done; done
echo "cfg5 xcd 0/1"
DESMAN_HIP_NTAB_XCD=0 run --V 50000 --S 96 --G 12 --steps 100 --warmup 20
DESMAN_HIP_NTAB_XCD=1 run --V 50000 --S 96 --G 12 --steps 100 --warmup 20
echo "cfg2 xcd 0/1"
DESMAN_HIP_NTAB_XCD=0 run --V 1000 --S 16 --G 5
DESMAN_HIP_NTAB_XCD=1 run --V 1000 --S 16 --G 5
