#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
( time python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -4 ) 2>&1 | tee $O/neartie2_pytest.txt
bash scripts/dbg/lib_ab.sh "10000 64 8" "50000 96 12" "20000 64 8" -- prev=desman_amd/lib/libdesman_hip_prev.so cur=desman_amd/lib/libdesman_hip.so 2>&1 | tee $O/r04_neartie_ab.txt
python scripts/dbg/chain_fp64.py 12 2>&1 | grep -E "share|burn-in|sampling|NMF" | tee $O/r04_chain_fp64.txt
python scripts/dbg/chain_fp64.py 8 2>&1 | grep -E "share|burn-in|sampling|NMF" | tee -a $O/r04_chain_fp64.txt
python scripts/dbg/chain_fp64.py 10 2>&1 | grep -E "share|burn-in|sampling|NMF" | tee -a $O/r04_chain_fp64.txt
