#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo skip generative tests

timeout -k 10 400 python -m pytest tests/test_blocks_gpu.py -x -q -s -k "sparse_and_hier" > gpurun_out/t40b.log 2>&1 || { tail -40 gpurun_out/t40b.log; exit 1; }
grep "moe variant" gpurun_out/t40b.log; tail -2 gpurun_out/t40b.log
timeout -k 10 400 python scratch/gen_bench.py 32 > gpurun_out/gen40.log 2>&1 || { tail -20 gpurun_out/gen40.log; exit 1; }
grep generative gpurun_out/gen40.log
