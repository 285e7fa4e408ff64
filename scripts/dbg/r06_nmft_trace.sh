#!/bin/bash
# round 6: rocprofv3 --kernel-trace --stats of the NMF update at 50k x 96 x {4, 8, 12} (scripts/prof_nmft.py: 5 + 300 + 50 updates, the three-launch loop)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_nmft_trace; mkdir -p $O
for G in 4 8 12; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$G -o t -- python scripts/prof_nmft.py 50000 96 $G 300 > $O/run_G$G.txt 2>&1
  f=$(ls $O/t$G/*/t_kernel_stats.csv $O/t$G/t_kernel_stats.csv 2>/dev/null | head -1)
  { echo "# 50k x 96 x $G: $(tail -1 $O/run_G$G.txt)"; head -8 "$f"; } >> $O/r06_nmft_kernel_stats.txt
  rm -rf $O/t$G
done
cat $O/r06_nmft_kernel_stats.txt
