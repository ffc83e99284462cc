#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/s2_krot_suite.log 2>&1; echo "suite rc=$?"; grep "FAILED\|passed\|failed" gpurun_out/s2_krot_suite.log | cut -c1-200 | tail -12
for i in 1 2; do
  for v in 0 1; do
    timeout -k 10 300 python bench.py --gemm-k-rotate $v --no-cpu-baseline --no-second-workload > gpurun_out/s2_ab_krot${v}_$i.log 2>&1 || { tail -5 gpurun_out/s2_ab_krot${v}_$i.log; exit 1; }
    python - <<P
import json
l=json.loads(open('gpurun_out/s2_ab_krot${v}_$i.log').read().strip().split('\n')[-1])
print('train-mode k-rotate $v run $i', l['ms_per_step'], 'gemm_ms', l['roofline']['gemm_ms_per_step'], 'frac', l['roofline']['frac'])
P
  done
done
