#!/bin/bash
# Round 3, runs ON the GPU box (via gpurun): kernel-trace stats of the bench command, `bench.py --pmc` (its own three PMC passes) at the
# single-GPU BASELINE shapes, the NMFT kernels under PMC (MFMA-pipe counters), and the bench lines themselves.  Output: gpurun_out/r03/.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --batch 0 > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $O/trace/*/t_kernel_stats.csv $O/r03_kernel_stats.csv 2>/dev/null || cp $O/trace/t_kernel_stats.csv $O/r03_kernel_stats.csv
# the benchmark line with traffic measured by the run itself, configs 3 (headline), 2 and the config-5 shape
timeout 400 python bench.py --steps 500 --warmup 50 --pmc > $O/r03_bench.json 2> $O/bench.err
timeout 300 python bench.py --V 1000 --S 16 --G 5 --steps 500 --warmup 50 --pmc --no-cpu-baseline > $O/r03_bench_cfg2_V1k_S16_G5.json 2>> $O/bench.err
timeout 400 python bench.py --V 50000 --S 96 --G 12 --steps 100 --warmup 20 --pmc --no-cpu-baseline > $O/r03_bench_cfg5_V50k_S96_G12.json 2>> $O/bench.err
cp gpurun_out/pmc_traffic_by_shape.json $O/pmc_traffic_by_shape.json
# NMFT: wall time per update + MFMA / VALU counters of the persistent kernel and of the three-launch kernels
python scripts/prof_nmft.py 10000 64 8 1000 > $O/r03_nmft.txt 2>&1
DESMAN_HIP_NMFT_NO_PERSIST=1 python scripts/prof_nmft.py 10000 64 8 1000 >> $O/r03_nmft.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_nmft -o p -- python scripts/prof_nmft.py 10000 64 8 200 > $O/pmc_nmft.log 2>&1
python scripts/summarize_pmc.py $O/r03_nmft_pmc_V10k.csv $O/pmc_nmft
python scripts/dbg/nmft_ab.py > $O/r03_nmft_ab.txt 2>&1
rm -rf $O/trace $O/pmc_nmft
ls -la $O
