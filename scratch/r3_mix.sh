#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
bash scratch/dw_lab.sh 2>&1 | tee $O/r3_dw_lab.log
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -x -k "fused_adamw" 2>&1 | tail -3
timeout -k 10 900 python scratch/krot_realizations.py > $O/r3_krot_real.log 2>&1; grep "^REAL" $O/r3_krot_real.log
