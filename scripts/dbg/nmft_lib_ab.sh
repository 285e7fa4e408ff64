#!/bin/bash
# A/B of library builds for the NMF start: nmft_lib_ab.sh "V S G" ... -- name=path ...  (DESMAN_HIP_LIB selects the build); us per update, twice, interleaved
shapes=(); while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; break; fi; shapes+=("$1"); shift; done
libs=("$@")
for shp in "${shapes[@]}"; do for rep in 1 2; do for l in "${libs[@]}"; do name=${l%%=*}; path=${l#*=}
  echo -n "$name: "; DESMAN_HIP_LIB=$PWD/$path python scripts/prof_nmft.py $shp 300 2>&1 | tail -1
done; done; done
