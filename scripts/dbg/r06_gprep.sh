#!/bin/bash
# round 6: the shape-independent part of a gamma row's draws moved to stage 2's idle wavefront -- parity, then A/B against the build before
python -m pytest tests -m gpu -x -q -k "dirichlet or gibbs or stage2 or stats or chain or golden" 2>&1 | tail -3
{
for r in 1 2 3; do for lib in prev hip; do echo -n "$lib: "; DESMAN_HIP_LIB=$PWD/desman_amd/lib/libdesman_$( [ $lib = prev ] && echo hip_prev || echo hip ).so python bench.py --steps 500 --warmup 50 --no-cpu-baseline --batch 0 --no-pmc --no-nmft 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; print('%.2f us'%(d['ms_per_step']*1e3), {a: round(b,1) for a,b in k.items()})"; done; done
DESMAN_HIP_LIB=$PWD/desman_amd/lib/libdesman_hip_ab.so python scripts/dbg/r06_s2_clocks.py 2>&1 | tail -15
} 2>&1 | tee gpurun_out/r06_gprep.txt
