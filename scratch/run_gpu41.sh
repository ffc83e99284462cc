#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_kernels_gpu.py -x -q -k "cross_entropy or expert_row or gemm" > gpurun_out/t41a.log 2>&1 || { tail -40 gpurun_out/t41a.log; exit 1; }
tail -2 gpurun_out/t41a.log
timeout -k 10 700 python -m pytest tests/test_generative_gpu.py tests/test_blocks_gpu.py -x -q > gpurun_out/t41b.log 2>&1 || { tail -40 gpurun_out/t41b.log; exit 1; }
tail -2 gpurun_out/t41b.log
timeout -k 10 400 python scratch/gen_bench.py 32 > gpurun_out/gen41.log 2>&1 || { tail -20 gpurun_out/gen41.log; exit 1; }
grep generative gpurun_out/gen41.log
