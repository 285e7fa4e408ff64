#!/bin/bash
export DESMAN_HIP_LIB=$PWD/desman_amd/lib/libdesman_hip_ab.so
for r in 0 1; do for w in 6 5; do echo -n "REGG $r wgs $w: "; DESMAN_HIP_STATS_REGG=$r DESMAN_HIP_STATS_WGS=$w python scripts/dbg/r06_s1_clocks.py 2>&1 | grep -E "kernel span|prologue dt" | tr '\n' ' '; echo; done; done
for r in 0 1; do echo -n "bench REGG $r: "; DESMAN_HIP_STATS_REGG=$r python bench.py --steps 200 --warmup 30 --no-cpu-baseline --batch 0 --no-pmc --no-nmft 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; print('%.2f us'%(d['ms_per_step']*1e3), {a: round(b,1) for a,b in k.items()})"; done
