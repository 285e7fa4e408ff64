#!/bin/bash
# A/B of library builds on the GPU box: lib_ab.sh "V S G" ... -- name=path ...   (paths relative to the repo; DESMAN_HIP_LIB selects the build)
# prints ms per Gibbs iteration (median of 5 repeats of 200 steps) and the per-kernel event times for every shape x build, interleaved twice
shapes=(); libs=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; break; fi; shapes+=("$1"); shift; done
libs=("$@")
for shp in "${shapes[@]}"; do set -- $shp
for rep in 1 2; do for l in "${libs[@]}"; do name=${l%%=*}; path=${l#*=}
  echo -n "$shp $name: "; DESMAN_HIP_LIB=$PWD/$path python bench.py --V $1 --S $2 --G $3 --steps 200 --warmup 30 --no-cpu-baseline --batch 0 --no-pmc --no-nmft 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; r=d['ms_per_step_repeats']; print('%.2f us (min %.2f max %.2f)' % (r['median']*1e3, r['min']*1e3, r['max']*1e3), {a: round(b,1) for a,b in k.items()}, 'fp64 frac %.4f' % d['roofline']['tau_steps_fp64_frac'])"
done; done; done
