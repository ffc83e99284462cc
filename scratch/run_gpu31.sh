#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_generative_gpu.py tests/test_blocks_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r2_t31.log 2>&1
rc=$?; echo "generative/blocks tests rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r2_t31.log | tail -8 | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python scratch/gen_bench.py 32 > gpurun_out/r2_gen_bench3.log 2>&1; grep -E "^generative" gpurun_out/r2_gen_bench3.log | cut -c1-300
