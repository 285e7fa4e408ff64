#!/bin/bash
# round 6: three workgroups per CU (three wavefronts per SIMD, 168 registers) for the five / six-tile update kernel at G = 5..8 with the four-block
# contractions (operands fetched per tile, B operands behind the tile's barriers): five tiles fit without scratch, six spill four registers
{
L=$PWD/desman_amd/lib
for shape in "50000 80 8" "50000 80 5" "50000 96 8" "50000 96 5"; do
for lib in hip w3b hip w3b; do
echo -n "$lib  "; DESMAN_HIP_LIB=$L/libdesman_$( [ $lib = w3b ] && echo hip_w3b || echo hip ).so python scripts/prof_nmft.py $shape 300 2>&1 | tail -1
done; done
} 2>&1 | tee gpurun_out/r06_nmft_b4_w3.txt
