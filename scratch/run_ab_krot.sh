#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
  for v in 0 1; do
    timeout -k 10 300 python bench.py --gemm-k-rotate $v --no-cpu-baseline --no-second-workload > gpurun_out/s2_ab_krot${v}_$i.log 2>&1 || { tail -5 gpurun_out/s2_ab_krot${v}_$i.log; exit 1; }
    python - <<P
import json
l=json.loads(open('gpurun_out/s2_ab_krot${v}_$i.log').read().strip().split('\n')[-1])
print('k-rotate $v run $i', l['ms_per_step'], 'gemm_ms', l['roofline']['gemm_ms_per_step'], 'frac', l['roofline']['frac'], 'loss', l['config']['final_loss'])
P
  done
done
for v in 0 1; do
    timeout -k 10 300 python bench.py --gemm-k-rotate $v --workload cfg3_mcan_moe4 --no-cpu-baseline --no-second-workload --no-roofline > gpurun_out/s2_ab3_krot${v}.log 2>&1 || exit 1
    python -c "
import json; l=json.loads(open('gpurun_out/s2_ab3_krot${v}.log').read().strip().split('\n')[-1]); print('cfg3 k-rotate $v', l['ms_per_step'])"
done
