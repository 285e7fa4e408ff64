#!/bin/bash
# round 6, measured and not kept (the kernel is not in the tree: nmft_rgl_kernel -- nmft_reduce_kernel's sums, sixteen statistics per workgroup, stored at
# the memory side; a ticket; the workgroup that draws the last runs nmft_gamma_body on agent-scope loads): the reduce + gamma / control step of a large
# table's NMF update as two launches (NMFT_FUSED=0) against that one launch (NMFT_FUSED=2) and one launch of column workgroups (NMFT_FUSED=1)
python -m pytest tests -m gpu -x -q -k "nmft or factorize or nmf" 2>&1 | grep -E "passed|failed|error" | tail -3
{
for shape in "50000 96 8" "50000 96 12" "50000 96 4" "10000 64 8" "10000 192 8" "3000 64 5"; do
for f in 0 2 1 0 2; do
echo -n "fused=$f  "; NMFT_FUSED=$f python scripts/prof_nmft.py $shape 300 2>&1 | tail -1
done; done
} 2>&1 | tee gpurun_out/r06_nmft_rgl.txt
