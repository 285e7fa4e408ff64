#!/bin/bash
# run the Gibbs loop with the instrumented library (desman_amd/lib_prof) swapped in for the duration of the run
cp desman_amd/lib/libdesman_hip.so /tmp/lib_orig.so
cp desman_amd/lib_prof/libdesman_hip.so desman_amd/lib/libdesman_hip.so
for spec in 2 3; do
echo "spec $spec"
DESMAN_HIP_STATS_SPEC=$spec DESMAN_HIP_STATS_REGG=0 python bench.py --steps 600 --warmup 50 --no-cpu-baseline --batch 0 2>&1 | grep PROF | tail -3
done
cp /tmp/lib_orig.so desman_amd/lib/libdesman_hip.so
