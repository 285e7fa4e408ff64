#!/bin/bash
# Round 6, runs ON the GPU box (via gpurun): `pytest -m gpu`, the default bench line (its own PMC passes), kernel-trace stats of the bench
# command, the bench at the other single-GPU BASELINE shapes and the wide shapes, the NMF start and factorize_tau per shape, chain phases /
# cost components, the -r workflow, config 5 whole on one GPU.  Output: gpurun_out/r06/ (the r06_* files are copied to profiles/).
# usage: collect_profiles_r06.sh [quick]      quick = the bench lines and the kernel trace only
MODE=$1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
( time python bench.py --steps 20 --warmup 5 > $O/r06_bench_driver_default.json 2> $O/bench_default.err ) 2>&1 | tail -3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --batch 0 --no-pmc > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $O/trace/*/t_kernel_stats.csv $O/r06_kernel_stats.csv 2>/dev/null || cp $O/trace/t_kernel_stats.csv $O/r06_kernel_stats.csv
rm -rf $O/trace
timeout 600 python bench.py --steps 500 --warmup 50 > $O/r06_bench.json 2> $O/bench.err
for shp in "50000 96 12" "50000 96 4"; do set -- $shp
  timeout 600 python bench.py --V $1 --S $2 --G $3 --steps 100 --warmup 20 --no-cpu-baseline --batch 0 > $O/r06_bench_V$1_S$2_G$3.json 2>> $O/bench.err
done
cp gpurun_out/pmc_traffic_by_shape.json $O/pmc_traffic_by_shape.json
python scripts/kernel_regs.py > $O/r06_kernel_regs.txt 2>&1
[ "$MODE" = quick ] && { ls -la $O; exit 0; }
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) 2>&1 | tee $O/r06_pytest_gpu.txt
timeout 300 python bench.py --V 1000 --S 16 --G 5 --steps 500 --warmup 50 --no-cpu-baseline > $O/r06_bench_cfg2_V1k_S16_G5.json 2>> $O/bench.err
for shp in "10000 192 8" "10000 300 8" "5000 512 8"; do set -- $shp
  timeout 600 python bench.py --V $1 --S $2 --G $3 --steps 100 --warmup 20 --no-cpu-baseline --batch 0 > $O/r06_bench_V$1_S$2_G$3.json 2>> $O/bench.err
done
for shp in "1000 64 5" "3000 64 5" "10000 64 8" "30000 64 6" "50000 96 8" "50000 96 12"; do python scripts/dbg/prof_nmft_tau.py $shp 2>&1 | tail -1; done > $O/r06_nmft_factorize_tau.txt
python scripts/chain_phases.py --out $O/r06_chain_phases.json 2>&1 | grep "G=" > $O/r06_chain_phases.txt
python scripts/fit_chain_cost.py --out $O/r06_chain_cost_components.json > $O/r06_chain_cost_components.txt 2>&1
python scripts/bench_rpath.py --out $O/r06_rpath.json > $O/r06_rpath.log 2>&1
python scripts/bench_config5.py --modes one,threads4,batch2 --out $O/r06_config5.json > $O/r06_config5.log 2>&1
ls -la $O
