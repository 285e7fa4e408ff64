#!/bin/bash
# rocprofv3 kernel stats of the Gibbs loop at one shape: r06_trace_shape.sh V S G [trueG]
cd /tmp && export TMPDIR=/tmp
V=$1; S=$2; G=$3; TG=${4:-$3}
O=/tmp/tr_$V_$S_$G; rm -rf $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o t -- python $GRAFT_REPO_ROOT/bench.py --V $V --S $S --G $G --true-G $TG --steps 100 --warmup 100 --repeats 2 --no-cpu-baseline --batch 0 --no-pmc --no-nmft > /dev/null 2>&1
f=$(find $O -name "t_kernel_stats.csv" | head -1)
echo "== $V x $S x $G (table from $TG strains)"; head -14 $f | cut -d, -f1-4,6-7 | sed 's/(StatsAggParams)//; s/(TauParams)//'
