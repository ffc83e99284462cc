#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_dp_gpu.py -m gpu -q -p no:cacheprovider -k "causal or smoothing or expert_row or moe_dense" > gpurun_out/r2_t24.log 2>&1
rc=$?; echo "rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r2_t24.log | tail -14 | cut -c1-300
