#!/bin/bash
# round 6, measured and not kept (the change is not in the tree): the first quad's F tiles asked for before the update kernel stages its operands
# (five / six sample tiles) -- against the build before (libdesman_hip_prev.so)
python -m pytest tests -m gpu -x -q -k "nmft or factorize or nmf" 2>&1 | grep -E "passed|failed|rror" | tail -3
{
L=$PWD/desman_amd/lib
for shape in "50000 96 8" "50000 96 12" "50000 96 4" "50000 80 6" "20000 96 8"; do
for lib in prev hip prev hip; do
echo -n "$lib  "; DESMAN_HIP_LIB=$L/libdesman_$( [ $lib = prev ] && echo hip_prev || echo hip ).so python scripts/prof_nmft.py $shape 300 2>&1 | tail -1
done; done
echo "== gamma fixed"
for shape in "50000 96 8"; do
for lib in prev hip prev hip; do
echo -n "$lib  "; DESMAN_HIP_LIB=$L/libdesman_$( [ $lib = prev ] && echo hip_prev || echo hip ).so python scripts/dbg/prof_nmft_tau.py $shape 300 2>&1 | tail -1
done; done
} 2>&1 | tee gpurun_out/r06_nmft_ffirst.txt
