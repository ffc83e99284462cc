#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -x -k "grouped_weight" 2>&1 | tail -3
timeout -k 10 300 python scratch/dw256_bench.py 2>&1 | tail -2 | sed 's/^/stagger: /'
VQA_HIP_LIB=$R/scratch/libvqa_dwnostag.so timeout -k 10 300 python scratch/dw256_bench.py 2>&1 | tail -2 | sed 's/^/nostagger: /'
VQA_HIP_LIB=$R/scratch/libvqa_dwtrace.so DW_MODE=big timeout -k 10 200 python scratch/dw_trace.py 2>/dev/null
