#!/bin/bash
# round 6: stage 1 of the mu/E pass, builds side by side (GPU box).  Output: gpurun_out/r06_stats_ab.txt
mkdir -p gpurun_out
{
echo "== parity (stats / gibbs legs of the GPU suite) with the new product library"
timeout 1500 python -m pytest tests -m gpu -x -q -k "stats or gibbs or batch or law or spec" 2>&1 | tail -5
echo "== A/B"
bash scripts/dbg/lib_ab.sh "10000 64 8" "50000 96 12" "20000 32 5" "10000 96 8" -- r5=desman_amd/lib/libdesman_hip_r5.so new=desman_amd/lib/libdesman_hip.so nopf=desman_amd/lib/libdesman_hip_nopf.so ereg=desman_amd/lib/libdesman_hip_ereg.so
} > gpurun_out/r06_stats_ab.txt 2>&1
tail -40 gpurun_out/r06_stats_ab.txt
