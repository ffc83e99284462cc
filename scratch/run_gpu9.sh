#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_dp_gpu.py tests/test_graph_gpu.py tests/test_blocks_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r2_t9.log 2>&1
rc=$?; echo "pytest rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r2_t9.log | tail -8 | cut -c1-300
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
for i in 1 2; do for v in main noprefetch; do
  if [ $v = main ]; then unset VQA_HIP_LIB; else export VQA_HIP_LIB=$PWD/scratch/libvqa_$v.so; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-second-workload > gpurun_out/r2_ab_$v$i.log 2> gpurun_out/r2_ab_$v$i.err || exit 1
  python - <<PY
import json
l=[x for x in open('gpurun_out/r2_ab_$v$i.log') if x.startswith('{')][-1]; d=json.loads(l)
print('$v$i', 'ms', d['ms_per_step'], 'gemm', d['roofline']['gemm_ms_per_step'], 'frac', d['roofline']['frac'])
PY
done; done
