export DESMAN_HIP_LIB=${DESMAN_HIP_LIB:-$PWD/desman_amd/lib/libdesman_hip_ab.so}   # the experiment build: A/B switches compiled in (make -C desman_amd/csrc ab)
for i in 1 2 3; do
for t in 1 0; do echo -n "tune=$t: "; DESMAN_HIP_NTAB_TUNE=$t python bench.py --steps 300 --warmup 30 --no-cpu-baseline --batch 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; print(round(d['ms_per_step']*1000,1), {a:round(b,1) for a,b in k.items() if a!='mt'})"; done; done
