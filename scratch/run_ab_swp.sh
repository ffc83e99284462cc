#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or grouped or fused_inproj" > gpurun_out/s2_swp_tests.log 2>&1 || { tail -30 gpurun_out/s2_swp_tests.log; exit 1; }
tail -1 gpurun_out/s2_swp_tests.log
for v in base swp; do
  if [ $v = swp ]; then export VQA_HIP_LIB=$PWD/scratch/libvqa_swp.so; else unset VQA_HIP_LIB; fi   # scratch/ab_build.sh swp -DVQA_GEMM_SWP=1
  echo "== $v"
  timeout -k 10 200 python scratch/group_dw_bench.py 2048 2>&1 | grep "default\|256x128/3\|128x128/3"
  timeout -k 10 200 python scratch/gemm_cold_tiles.py 2>&1 | grep -v amdgpu | cut -c1-75
done
for i in 1 2; do
  for v in base swp; do
    if [ $v = swp ]; then export VQA_HIP_LIB=$PWD/scratch/libvqa_swp.so; else unset VQA_HIP_LIB; fi   # scratch/ab_build.sh swp -DVQA_GEMM_SWP=1
    timeout -k 10 300 python bench.py --no-second-workload --no-cpu-baseline > gpurun_out/s2_ab_${v}_$i.log 2>&1 || exit 1
    python - <<P
import json
l=json.loads(open('gpurun_out/s2_ab_${v}_$i.log').read().strip().split('\n')[-1])
print('$v $i', l['ms_per_step'], 'gemm_ms', l['roofline']['gemm_ms_per_step'], 'frac', l['roofline']['frac'], 'fusion', l['roofline']['fusion_mfma_util'])
P
  done
done
