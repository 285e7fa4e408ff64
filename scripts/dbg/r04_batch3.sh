#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
( time python -m pytest tests/test_gpu_genes.py tests/test_gpu_host.py -x -q 2>&1 | tail -5 ) 2>&1 | tee $O/genes_host_pytest.txt
python scripts/misfit_scan.py --out $O/r04_misfit_scan.json 2>&1 | tee $O/r04_misfit_scan.txt
python scripts/misfit_scan.py --V 10000 --S 64 --true-G 4 --gs 2,3,4,6,8 --out $O/r04_misfit_scan_10k.json 2>&1 | tee -a $O/r04_misfit_scan.txt
echo "== concurrent chains (own streams)"; for k in 2 4; do for one in 0 1; do
  echo -n "chains-per-gpu $k ONE_STREAM=$one: "; DESMAN_HIP_ONE_STREAM=$one python bench.py --steps 200 --warmup 30 --no-pmc --no-cpu-baseline --batch 0 --no-nmft --chains-per-gpu $k 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['chains_per_gpu'])"
done; done 2>&1 | tee $O/r04_concurrent.txt
