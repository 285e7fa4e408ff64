#!/bin/bash
# round 6: at nine to twelve haplotypes only the GAMMA numerators on the four-block instruction (operands held in registers where there are
# 256: no extra fetches per tile) -- build with EXTRA=-DNM_B4G_MAXKB=3 LIBNAME=libdesman_hip_g3.so
{
L=$PWD/desman_amd/lib
for shape in "50000 96 12" "50000 96 9" "50000 80 10" "10000 64 12" "10000 192 12"; do
for lib in hip g3 hip g3; do
echo -n "$lib  "; DESMAN_HIP_LIB=$L/libdesman_$( [ $lib = g3 ] && echo hip_g3 || echo hip ).so python scripts/prof_nmft.py $shape 300 2>&1 | tail -1
done; done
} 2>&1 | tee gpurun_out/r06_nmft_b4_g3.txt
