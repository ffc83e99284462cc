#!/bin/bash
# round-2 GPU call: whole GPU test suite, then the benchmark in both operand modes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=10 -p no:cacheprovider > gpurun_out/r2_t1.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r2_t1.log
tail -5 gpurun_out/r2_t1.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 600 python bench.py > gpurun_out/r2_b1.log 2> gpurun_out/r2_b1.err
rc=$?
echo "bench rc=$rc"; tail -c 2500 gpurun_out/r2_b1.log
if [ $rc -ne 0 ]; then tail -20 gpurun_out/r2_b1.err; exit $rc; fi
timeout -k 10 400 python bench.py --dtype fp16 --no-cpu-baseline --steps 50 --warmup 10 > gpurun_out/r2_b1_f16.log 2> gpurun_out/r2_b1_f16.err
echo "bench fp16 rc=$?"; tail -c 1500 gpurun_out/r2_b1_f16.log; tail -5 gpurun_out/r2_b1_f16.err
