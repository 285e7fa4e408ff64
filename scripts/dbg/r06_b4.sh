#!/bin/bash
# round 6: the haplotype-indexed contractions of the NMF update on v_mfma_f64_4x4x4_4b_f64 (kernels_nmft.hip: nm_b4) -- us per update
# against the build before (git stash; make LIBNAME=libdesman_hip_prev.so OBJDIR=../lib/obj_prev) at the shapes of VERDICT r5 item 2, on the
# small-table paths and with gamma fixed; then three workgroups per CU where the smaller allocation at G <= 4 allows it
# (make EXTRA=-DNM_WGS3_KB1 LIBNAME=libdesman_hip_w3.so OBJDIR=../lib/obj_w3)
{
L=$PWD/desman_amd/lib
for shape in "50000 96 8" "50000 96 5" "50000 96 4" "50000 96 2" "50000 96 12" "50000 96 16" "10000 64 8" "10000 192 8" "5000 512 8" "20000 32 6" "3000 64 5" "1000 64 5"; do
for lib in prev hip prev hip; do
echo -n "$lib  "; DESMAN_HIP_LIB=$L/libdesman_$( [ $lib = prev ] && echo hip_prev || echo hip ).so python scripts/prof_nmft.py $shape 300 2>&1 | tail -1
done; done
echo "== gamma fixed (factorize_tau)"
for shape in "50000 96 8" "50000 96 4" "10000 64 8" "1000 64 5"; do
for lib in prev hip; do
echo -n "$lib  "; DESMAN_HIP_LIB=$L/libdesman_$( [ $lib = prev ] && echo hip_prev || echo hip ).so python scripts/dbg/prof_nmft_tau.py $shape 300 2>&1 | tail -1
done; done
echo "== three workgroups per CU at G <= 4, five / six tiles (w3) against two (hip)"
for shape in "50000 96 4" "50000 96 2" "50000 80 4"; do
for lib in hip w3 hip w3; do
echo -n "$lib  "; DESMAN_HIP_LIB=$L/libdesman_$( [ $lib = w3 ] && echo hip_w3 || echo hip ).so python scripts/prof_nmft.py $shape 300 2>&1 | tail -1
done; done
} 2>&1 | tee gpurun_out/r06_nmft_b4.txt
