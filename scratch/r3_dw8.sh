#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
VQA_HIP_LIB=$R/scratch/libvqa_dwfirst.so timeout -k 10 300 python scratch/dw256_bench.py 2>&1 | tail -2 | sed 's/^/dma first: /'
VQA_HIP_LIB=$R/scratch/libvqa_dwnopf.so timeout -k 10 300 python scratch/dw256_bench.py 2>&1 | tail -2 | sed 's/^/dma behind reads: /'
VQA_HIP_LIB=$R/scratch/libvqa_dwfirsttr.so DW_MODE=mix timeout -k 10 200 python scratch/dw_trace.py 2>/dev/null | sed 's/^/dma first, last tile: /' | tail -6
