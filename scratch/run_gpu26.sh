#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_generative_gpu.py -m gpu -q -p no:cacheprovider -s  > gpurun_out/r2_t26.log 2>&1
rc=$?; echo "fusion tests rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  |FUSION" gpurun_out/r2_t26.log | tail -12 | cut -c1-400
