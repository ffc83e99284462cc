#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -x -q > gpurun_out/s2_mask_tests.log 2>&1 || { tail -40 gpurun_out/s2_mask_tests.log; exit 1; }
tail -1 gpurun_out/s2_mask_tests.log
for i in 1 2; do
  for v in prev head; do
    if [ $v = head ]; then unset VQA_HIP_LIB; else export VQA_HIP_LIB=$PWD/scratch/libvqa_$v.so; fi
    timeout -k 10 300 python bench.py --no-second-workload --no-cpu-baseline --no-roofline > gpurun_out/s2_ab_${v}_$i.log 2>&1 || exit 1
    python -c "
import json; l=json.loads(open('gpurun_out/s2_ab_${v}_$i.log').read().strip().split('\n')[-1]); print('$v $i', l['ms_per_step'])"
  done
done
unset VQA_HIP_LIB
bash scratch/trace_by_shape.sh 2>&1 | grep "fused_inproj\|attn_mfma\|total kernel"
