#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > gpurun_out/t39a.log 2>&1 || { tail -30 gpurun_out/t39a.log; exit 1; }
tail -2 gpurun_out/t39a.log
timeout -k 10 600 python -m pytest tests/test_generative_gpu.py -x -q > gpurun_out/t39b.log 2>&1 || { tail -30 gpurun_out/t39b.log; exit 1; }
tail -2 gpurun_out/t39b.log
timeout -k 10 400 python scratch/gen_bench.py 32 > gpurun_out/gen39.log 2>&1 || { tail -20 gpurun_out/gen39.log; exit 1; }
grep generative gpurun_out/gen39.log
