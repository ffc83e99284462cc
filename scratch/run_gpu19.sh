#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_generative_gpu.py -m gpu -q -p no:cacheprovider -s -x > gpurun_out/r2_t19.log 2>&1
rc=$?; echo "generative tests rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  |GENERATIVE" gpurun_out/r2_t19.log | tail -16 | cut -c1-400
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "attention or cross_entropy" > gpurun_out/r2_t19b.log 2>&1
rc=$?; echo "kernel tests rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r2_t19b.log | tail -6 | cut -c1-300
