#!/bin/bash
# round 6: the gamma / control step of a large table's NMF update at the start of the update kernel (NMFT_FUSED=3; the default above 128
# partials on the matrix-core path) against a launch of its own (NMFT_FUSED=0) and one launch with the reduction (NMFT_FUSED=1)
python -m pytest tests -m gpu -x -q -k "nmft or factorize or nmf" 2>&1 | grep -E "passed|failed|rror" | tail -5
{
for shape in "50000 96 8" "50000 96 12" "50000 96 4" "10000 64 8" "20000 32 6" "3000 64 5"; do
for f in 0 3 0 3; do
echo -n "fused=$f  "; NMFT_FUSED=$f python scripts/prof_nmft.py $shape 300 2>&1 | tail -1
done; done
} 2>&1 | tee gpurun_out/r06_nmft_gstep.txt
