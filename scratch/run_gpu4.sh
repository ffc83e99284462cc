#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for m in fp16 bf16; do timeout -k 10 200 python scratch/dbg_fp16_head.py tiny_concat $m 1024 2>&1 | grep -v Warn | tail -8; done
for ws in 0 1; do
timeout -k 10 400 python bench.py --no-cpu-baseline --no-second-workload --gemm-ws $ws > gpurun_out/r2_b4_ws$ws.log 2> gpurun_out/r2_b4_ws$ws.err; echo "bench ws=$ws rc=$?"; python - <<PY
import json
l=[x for x in open('gpurun_out/r2_b4_ws$ws.log') if x.startswith('{')][-1]; d=json.loads(l)
print('ms', d['ms_per_step'], 'gemm', d['roofline']['gemm_ms_per_step'], d['roofline']['launches_per_step'], 'frac', d['roofline']['frac'], 'fusion', d['roofline']['fusion_mfma_util'], d['roofline']['gemm_gflop_per_step'])
PY
done
