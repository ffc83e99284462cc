#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
  for v in plain asm; do
    if [ $v = plain ]; then export VQA_HIP_LIB=$PWD/scratch/libvqa_plain.so; else unset VQA_HIP_LIB; fi
    timeout -k 10 300 python bench.py --no-second-workload --no-cpu-baseline > gpurun_out/s2_ab_${v}_$i.log 2>&1 || exit 1
    python - <<P
import json
l=json.loads(open('gpurun_out/s2_ab_${v}_$i.log').read().strip().split('\n')[-1])
print('$v $i', l['ms_per_step'], 'gemm_ms', l['roofline']['gemm_ms_per_step'], 'frac', l['roofline']['frac'], 'fusion', l['roofline']['fusion_mfma_util'])
P
  done
done
