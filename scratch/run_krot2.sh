#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py tests/test_parity_gpu.py -x -q > gpurun_out/s2_krot_tests.log 2>&1 || { tail -40 gpurun_out/s2_krot_tests.log; exit 1; }
tail -1 gpurun_out/s2_krot_tests.log
for i in 1 2; do
  for v in 1 2; do
    timeout -k 10 300 python bench.py --gemm-k-rotate $v --no-cpu-baseline --no-second-workload > gpurun_out/s2_ab_krot${v}_$i.log 2>&1 || { tail -5 gpurun_out/s2_ab_krot${v}_$i.log; exit 1; }
    python - <<P
import json
l=json.loads(open('gpurun_out/s2_ab_krot${v}_$i.log').read().strip().split('\n')[-1])
print('k-rotate $v run $i', l['ms_per_step'], 'gemm_ms', l['roofline']['gemm_ms_per_step'], 'frac', l['roofline']['frac'])
P
  done
done
