#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -x -q > gpurun_out/s2_epi_tests.log 2>&1 || { tail -40 gpurun_out/s2_epi_tests.log; exit 1; }
tail -1 gpurun_out/s2_epi_tests.log
for v in prev noepi head; do
  if [ $v = head ]; then unset VQA_HIP_LIB; else export VQA_HIP_LIB=$PWD/scratch/libvqa_$v.so; fi
  echo "== $v"; timeout -k 10 200 python scratch/gemm_epi_bench.py 2>&1 | grep -v amdgpu
done
for i in 1 2; do
  for v in prev noepi head; do
    if [ $v = head ]; then unset VQA_HIP_LIB; else export VQA_HIP_LIB=$PWD/scratch/libvqa_$v.so; fi
    timeout -k 10 300 python bench.py --no-second-workload --no-cpu-baseline > gpurun_out/s2_ab_${v}_$i.log 2>&1 || exit 1
    python - <<P
import json
l=json.loads(open('gpurun_out/s2_ab_${v}_$i.log').read().strip().split('\n')[-1])
print('$v $i', l['ms_per_step'], 'gemm_ms', l['roofline']['gemm_ms_per_step'], 'frac', l['roofline']['frac'], 'fusion', l['roofline']['fusion_mfma_util'])
P
  done
done
