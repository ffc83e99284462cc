#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_generative_gpu.py tests/test_blocks_gpu.py -q -p no:cacheprovider -k "expert_row or linear_cross or generative or expert or moe or cross_entropy" > $O/gen_check.log 2>&1; tail -4 $O/gen_check.log
timeout -k 10 300 python scratch/gen_bench.py 2>&1 | grep -v amdgpu | tail -3
