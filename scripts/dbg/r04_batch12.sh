#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -4 ) 2>&1 | tee $O/spec4b_pytest.txt
python scripts/misfit_scan.py --gs 2,3,4,5,6 --out $O/r04_misfit_scan_spec4.json 2>&1 | tee $O/r04_misfit_scan_spec4c.txt
python scripts/misfit_scan.py --gs 7,8 --stats-spec 4 --out $O/r04_misfit_scan_spec4f.json 2>&1 | tee -a $O/r04_misfit_scan_spec4c.txt
python scripts/misfit_scan.py --V 10000 --S 64 --true-G 4 --gs 2,3,4,5 --stats-spec 4 --out $O/r04_misfit_scan_spec4_10k.json 2>&1 | tee -a $O/r04_misfit_scan_spec4c.txt
python scripts/misfit_scan.py --V 20000 --S 64 --true-G 4 --gs 2,3,4,5 --stats-spec 4 --out $O/r04_misfit_scan_spec4_20k.json 2>&1 | tee -a $O/r04_misfit_scan_spec4c.txt
python scripts/misfit_scan.py --V 20000 --S 64 --true-G 4 --gs 2,3,4,5 --stats-spec 2 --out $O/r04_misfit_scan_spec2_20k.json 2>&1 | tee -a $O/r04_misfit_scan_spec4c.txt
