#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" > gpurun_out/s2_order_tests.log 2>&1 || { tail -40 gpurun_out/s2_order_tests.log; exit 1; }
tail -1 gpurun_out/s2_order_tests.log
for i in 1 2; do
  for v in row col; do
    fl=""; [ $v = row ] && fl="--gemm-tile-order 1"
    timeout -k 10 300 python bench.py $fl --no-cpu-baseline --no-second-workload > gpurun_out/s2_ab_${v}_$i.log 2>&1 || { tail -5 gpurun_out/s2_ab_${v}_$i.log; exit 1; }
    python - <<P
import json
l=json.loads(open('gpurun_out/s2_ab_${v}_$i.log').read().strip().split('\n')[-1])
print('$v $i', l['ms_per_step'], 'gemm_ms', l['roofline']['gemm_ms_per_step'], 'frac', l['roofline']['frac'], 'fusion', l['roofline']['fusion_mfma_util'])
P
  done
done
for v in row col; do
    fl=""; [ $v = row ] && fl="--gemm-tile-order 1"
    timeout -k 10 300 python bench.py $fl --workload cfg3_mcan_moe4 --no-cpu-baseline --no-second-workload --no-roofline > gpurun_out/s2_ab3_${v}.log 2>&1 || exit 1
    python -c "
import json; l=json.loads(open('gpurun_out/s2_ab3_${v}.log').read().strip().split('\n')[-1]); print('cfg3 $v', l['ms_per_step'])"
done
