#!/bin/bash
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_gen
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_gen --output-format csv -- python3 $R/scratch/gen_bench.py 32 > $R/gpurun_out/prof_gen.log 2>&1
f=$(find $R/gpurun_out/prof_gen -name '*kernel_stats.csv' | head -1); cp $f $R/gpurun_out/prof_gen_kernel_stats.csv; head -25 $f | cut -c1-160
