#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_gen
export TMPDIR=/tmp
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_gen -o gen --output-format csv -- python3 $GRAFT_REPO_ROOT/scratch/gen_bench.py 32 > $GRAFT_REPO_ROOT/gpurun_out/gen_prof.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/gen_prof.log; exit 1; }
cd $GRAFT_REPO_ROOT
grep generative gpurun_out/gen_prof.log
ls -la gpurun_out/prof_gen | head
# keep only stats (trace is large)
find gpurun_out/prof_gen -name "*.db" -delete
