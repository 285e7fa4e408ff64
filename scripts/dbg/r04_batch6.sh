#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
( time python -m pytest tests/test_gpu_parity.py -x -q -k "stats or gibbs_update_is_self or law or recovers or posterior or per_read" 2>&1 | tail -15 ) 2>&1 | tee $O/spec4_pytest.txt
echo "== spec by rule (4 where 4 x 4^G <= V)"; python scripts/misfit_scan.py --gs 2,3,4,5,6 --out $O/r04_misfit_scan_spec4.json 2>&1 | tee $O/r04_misfit_scan_spec4.txt
echo "== forced spec 2"; python scripts/misfit_scan.py --gs 2,3,4,6 --stats-spec 2 --out $O/r04_misfit_scan_spec2.json 2>&1 | tee -a $O/r04_misfit_scan_spec4.txt
echo "== forced spec 4 beyond the rule"; python scripts/misfit_scan.py --gs 7,8 --stats-spec 4 --out $O/r04_misfit_scan_spec4f.json 2>&1 | tee -a $O/r04_misfit_scan_spec4.txt
echo "== 10k x 64"; python scripts/misfit_scan.py --V 10000 --S 64 --true-G 4 --gs 2,3,4,5 --out $O/r04_misfit_scan_spec4_10k.json 2>&1 | tee -a $O/r04_misfit_scan_spec4.txt
python scripts/misfit_scan.py --V 10000 --S 64 --true-G 4 --gs 2,3,4,5 --stats-spec 2 --out $O/r04_misfit_scan_spec2_10k.json 2>&1 | tee -a $O/r04_misfit_scan_spec4.txt
