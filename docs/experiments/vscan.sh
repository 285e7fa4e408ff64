export DESMAN_HIP_LIB=${DESMAN_HIP_LIB:-$PWD/desman_amd/lib/libdesman_hip_ab.so}   # the experiment build: A/B switches compiled in (make -C desman_amd/csrc ab)
for V in 3072 6144 7000 8000 9216 10000 12288 18432; do
python bench.py --V $V --S 64 --G 8 --steps 200 --warmup 20 --no-cpu-baseline --batch 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; print($V, round(d['ms_per_step']*1000,1), {a:round(b,1) for a,b in k.items() if a!='mt'}, d['roofline'].get('tau_launch'))"
done
