#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
DW_ZERO=1 timeout -k 10 300 python scratch/dw256_bench.py 2>&1 | tail -2 | sed 's/^/zero-data stagger: /'
VQA_HIP_LIB=$R/scratch/libvqa_dwnostag.so DW_ZERO=1 timeout -k 10 300 python scratch/dw256_bench.py 2>&1 | tail -2 | sed 's/^/zero-data nostagger: /'
VQA_HIP_LIB=$R/scratch/libvqa_dwtrace.so DW_MODE=big timeout -k 10 200 python scratch/dw_trace.py 2>/dev/null | tail -4
VQA_HIP_LIB=$R/scratch/libvqa_dwtracens.so DW_MODE=big timeout -k 10 200 python scratch/dw_trace.py 2>/dev/null | tail -4
