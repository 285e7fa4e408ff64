#!/bin/bash
# Round 4, runs ON the GPU box (via gpurun): the whole `pytest -m gpu` suite, the default bench line (its own PMC passes), kernel-trace
# stats of the bench command, the bench at the other single-GPU BASELINE shapes + the wide-S shapes of the lean tau sweep, the in-library
# RCCL exchange with a world of one, chain phases / cost components, what chains that fit badly cost, config 5 whole on one GPU.
# Output: gpurun_out/r04/ (the r04_* files are copied to profiles/).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) 2>&1 | tee $O/r04_pytest_gpu.txt
( time python bench.py --steps 20 --warmup 5 > $O/r04_bench_driver_default.json 2> $O/bench_default.err ) 2>&1 | tail -3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --batch 0 --no-pmc > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $O/trace/*/t_kernel_stats.csv $O/r04_kernel_stats.csv 2>/dev/null || cp $O/trace/t_kernel_stats.csv $O/r04_kernel_stats.csv
timeout 600 python bench.py --steps 500 --warmup 50 > $O/r04_bench.json 2> $O/bench.err
timeout 300 python bench.py --V 1000 --S 16 --G 5 --steps 500 --warmup 50 --no-cpu-baseline > $O/r04_bench_cfg2_V1k_S16_G5.json 2>> $O/bench.err
for shp in "50000 96 12" "10000 192 8" "10000 300 8" "5000 512 8"; do set -- $shp
  timeout 600 python bench.py --V $1 --S $2 --G $3 --steps 100 --warmup 20 --no-cpu-baseline --batch 0 > $O/r04_bench_V$1_S$2_G$3.json 2>> $O/bench.err
done
timeout 600 python bench.py --V 50000 --S 96 --G 4 --steps 100 --warmup 20 --no-cpu-baseline --batch 0 > $O/r04_bench_V50000_S96_G4.json 2>> $O/bench.err
cp gpurun_out/pmc_traffic_by_shape.json $O/pmc_traffic_by_shape.json
for shp in "50000 96 12" "200000 64 8" "10000 64 8"; do python scripts/bench_vshard_comm.py $shp 100; done > $O/r04_vshard_comm.txt 2>&1
for shp in "1000 64 5" "3000 64 5" "10000 64 8" "30000 64 6" "50000 96 8" "50000 96 12"; do python scripts/dbg/prof_nmft_tau.py $shp 2>&1 | tail -1; done > $O/r04_nmft_factorize_tau.txt
python scripts/chain_phases.py --out $O/r04_chain_phases.json 2>&1 | grep "G=" > $O/r04_chain_phases.txt
python scripts/fit_chain_cost.py --out $O/r04_chain_cost_components.json > $O/r04_chain_cost_components.txt 2>&1
python scripts/kernel_regs.py > $O/r04_kernel_regs.txt 2>&1
python scripts/misfit_scan.py --gs 2,3,4,5,6,7,8,10,12 --out $O/r04_misfit_scan.json > $O/r04_misfit_scan.txt 2>&1
python scripts/misfit_scan.py --V 10000 --S 64 --true-G 4 --gs 2,3,4,6,8 --out $O/r04_misfit_scan_10k.json >> $O/r04_misfit_scan.txt 2>&1
python scripts/bench_config5.py --modes one,threads4 --out $O/r04_config5.json > $O/r04_config5.log 2>&1
rm -rf $O/trace
ls -la $O
