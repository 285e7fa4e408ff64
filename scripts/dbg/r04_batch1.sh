#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
( time python -m pytest tests/test_gpu_fuzz.py -x -q 2>&1 | tail -5 ) 2>&1 | tee $O/fuzz_pytest.txt
for shp in "50000 96 12" "10000 192 8" "10000 300 8" "5000 512 8"; do set -- $shp
  timeout 600 python bench.py --V $1 --S $2 --G $3 --steps 100 --warmup 20 --pmc --no-cpu-baseline --batch 0 --no-nmft > $O/bench_pmc_V$1_S$2_G$3.json 2>> $O/bench_pmc.err
  python - $O/bench_pmc_V$1_S$2_G$3.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(d["config"]["workload"][:40], "ms/step %.4f" % d["ms_per_step"], {k: (round(v["avg_kernel_us"],1), v["algorithmic_bytes_per_launch"], v["traffic_bytes_pmc"]) for k,v in r["per_kernel"].items()})
PY
done
cp gpurun_out/pmc_traffic_by_shape.json $O/pmc_traffic_by_shape.json
echo "== concurrent chains (own streams)"; for k in 2 4; do for one in 0 1; do
  echo -n "chains-per-gpu $k ONE_STREAM=$one: "; DESMAN_HIP_ONE_STREAM=$one python bench.py --steps 200 --warmup 30 --no-pmc --no-cpu-baseline --batch 0 --no-nmft --chains-per-gpu $k 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['chains_per_gpu'])"
done; done
echo "== batch 4"; python bench.py --steps 200 --warmup 30 --no-pmc --no-cpu-baseline --batch 4 --no-nmft 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['batch'])"
python scripts/fit_chain_cost.py --out $O/chain_cost_components.json 2>&1 | tee $O/chain_cost_components.txt
