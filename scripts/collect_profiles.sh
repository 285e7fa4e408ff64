#!/bin/bash
# Runs ON the GPU box (via gpurun): kernel-trace stats of the bench command + PMC passes of a short
# Gibbs run, each counter group in its own pass; writes summaries under gpurun_out/$TAG/.
# Copy what should be judged into profiles/ afterwards (profiles/README.md lists the files).
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$TAG; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --batch 0 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $O/trace/t_kernel_stats.csv $O/${TAG}_kernel_stats.csv
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq -o p -- python scripts/prof_gibbs.py 30 > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $O/pmc_sq2 -o p -- python scripts/prof_gibbs.py 30 > $O/pmc_sq2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python scripts/prof_gibbs.py 30 > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python scripts/prof_gibbs.py 30 > $O/pmc_write.log 2>&1
python scripts/summarize_pmc.py $O/${TAG}_pmc.csv $O/pmc_sq $O/pmc_sq2 $O/pmc_fetch $O/pmc_write
python bench.py --steps 500 --warmup 50 > $O/${TAG}_bench.json 2> $O/bench.err
rm -rf $O/trace $O/pmc_sq $O/pmc_sq2 $O/pmc_fetch $O/pmc_write
ls -la $O
