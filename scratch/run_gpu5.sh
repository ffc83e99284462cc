#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 200 python scratch/dbg_fp16_bisect.py tiny_concat fp16 1024 2>&1 | grep -v Warn | tail -12
timeout -k 10 200 python scratch/lab_run_ws.py scratch/labws_64x288.so NT 2048 2304 768 > gpurun_out/r2_labws_64x288.log 2>&1; cat gpurun_out/r2_labws_64x288.log
timeout -k 10 200 python scratch/lab_run_ws.py scratch/labws_64x96.so NT 2048 768 3072 > gpurun_out/r2_labws_64x96.log 2>&1; cat gpurun_out/r2_labws_64x96.log | head -30
for ws in 1024 9216; do
timeout -k 10 400 python bench.py --no-cpu-baseline --no-second-workload --gemm-ws $ws > gpurun_out/r2_b5_ws$ws.log 2> gpurun_out/r2_b5_ws$ws.err; echo "bench ws=$ws rc=$?"; python - <<PY
import json
l=[x for x in open('gpurun_out/r2_b5_ws$ws.log') if x.startswith('{')][-1]; d=json.loads(l)
print('ms', d['ms_per_step'], 'gemm', d['roofline']['gemm_ms_per_step'], d['roofline']['launches_per_step'], 'frac', d['roofline']['frac'])
PY
done
