#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py tests/test_graph_gpu.py tests/test_parity_gpu.py -q > gpurun_out/s2_ln_tests.log 2>&1; echo "tests rc=$?"; grep "FAILED\|passed\|failed" gpurun_out/s2_ln_tests.log | cut -c1-200 | tail -6
for i in 1 2; do
  for v in prev head; do
    if [ $v = head ]; then unset VQA_HIP_LIB; else export VQA_HIP_LIB=$PWD/scratch/libvqa_$v.so; fi
    timeout -k 10 300 python bench.py --no-second-workload --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print('$v $i', l['ms_per_step'])"
  done
done
unset VQA_HIP_LIB
bash scratch/trace_by_shape.sh 2>&1 | grep "ln_fwd\|ln_bwd_kernel\|total kernel" | head -8
