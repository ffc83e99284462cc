#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python scratch/opt_overlap_bench.py > gpurun_out/opt_overlap.log 2> gpurun_out/opt_overlap.err || { tail -20 gpurun_out/opt_overlap.err; exit 1; }
cat gpurun_out/opt_overlap.log
