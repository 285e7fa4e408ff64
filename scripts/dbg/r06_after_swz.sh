#!/bin/bash
# round 6, after the row-map change: time line, what the table atomics still cost (ablation), per-XCD copies, workgroups per CU
export DESMAN_HIP_LIB=$PWD/desman_amd/lib/libdesman_hip_ab.so
{
echo "== time line"; python scripts/dbg/r06_s1_clocks.py 2>&1 | grep -E "kernel span|tables staged|load\+Gamma|four items|table atomics|before epilogue|^end"
echo "== no table atomics (dbg 36)"; DESMAN_HIP_STATS_DBG=36 python scripts/dbg/r06_s1_clocks.py 2>&1 | grep -E "kernel span"
echo "== no Esum adds (dbg 20)"; DESMAN_HIP_STATS_DBG=20 python scripts/dbg/r06_s1_clocks.py 2>&1 | grep -E "kernel span"
echo "== no draws (dbg 1)"; DESMAN_HIP_STATS_DBG=1 python scripts/dbg/r06_s1_clocks.py 2>&1 | grep -E "kernel span"
for x in 0 1; do for w in 5 6 7; do echo -n "xcd $x wgs $w: "; DESMAN_HIP_NTAB_XCD=$x DESMAN_HIP_STATS_WGS=$w python bench.py --steps 200 --warmup 30 --no-cpu-baseline --batch 0 --no-pmc --no-nmft 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; print('%.2f us'%(d['ms_per_step']*1e3), {a: round(b,1) for a,b in k.items()})"; done; done
} 2>&1 | tee gpurun_out/r06_after_swz.txt
