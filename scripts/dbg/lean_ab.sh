# A/B of the register-lean sweep (three samples per lane) against the fp64-prefix form.  The second library: kernels_gibbs.hip compiled with
# -DTAU_NO_LEAN and linked with the other objects into desman_amd/lib/libdesman_hip_nolean.so (not built by the Makefile).
export DESMAN_HIP_LIB=${DESMAN_HIP_LIB:-$PWD/desman_amd/lib/libdesman_hip_ab.so}   # the experiment build: A/B switches compiled in (make -C desman_amd/csrc ab)
L=desman_amd/lib
for shp in "50000 96 12" "50000 96 6" "20000 48 8" "10000 192 8"; do set -- $shp
for v in lean nolean lean nolean; do
  if [ $v = nolean ]; then cp $L/libdesman_hip.so /tmp/keep.so; cp $L/libdesman_hip_nolean.so $L/libdesman_hip.so; fi
  echo -n "$shp $v: "; python bench.py --V $1 --S $2 --G $3 --steps 100 --warmup 20 --no-cpu-baseline --batch 0 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_us']; print(round(d['ms_per_step']*1000,2), 'tau', round(k['tau'],1))"
  if [ $v = nolean ]; then cp /tmp/keep.so $L/libdesman_hip.so; fi
done; done
