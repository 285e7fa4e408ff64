#!/bin/bash
# round 6: up to four haplotypes at five / six sample tiles -- the vector-ALU contractions of round 5 (VC) against the four-block matrix
# instruction there too (the run that decided it: `hip` = the commit with both forms, fe7dc6a~, built with the vector form; `novc` = the same source built
# with -DNM_NO_VC.  The vector form and the switch are gone since; the script is kept as the record of how profiles/r06_nmft_b4_novc.txt was made)
{
for shape in "50000 96 4" "50000 96 3" "50000 96 2" "50000 80 4"; do
for lib in hip novc hip novc; do
f=$PWD/desman_amd/lib/libdesman_hip.so; [ $lib = novc ] && f=$PWD/desman_amd/lib/libdesman_hip_novc.so
echo -n "$lib  "; DESMAN_HIP_LIB=$f python scripts/prof_nmft.py $shape 300 2>&1 | tail -1
done; done
} 2>&1 | tee gpurun_out/r06_b4_novc.txt
