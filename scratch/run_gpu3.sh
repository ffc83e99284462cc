#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python scratch/dbg_fp16_concat.py tiny_concat fp16 1024 > gpurun_out/r2_dbg1.log 2>&1; echo "dbg rc=$?"; tail -40 gpurun_out/r2_dbg1.log
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "gemm" > gpurun_out/r2_t3.log 2>&1
rc=$?; echo "gemm tests rc=$rc"; tail -8 gpurun_out/r2_t3.log | cut -c1-250
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 600 python scratch/gemm_shapes_bench.py > gpurun_out/r2_gemm_shapes.log 2>&1; echo "shapes rc=$?"; cat gpurun_out/r2_gemm_shapes.log
