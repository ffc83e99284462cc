#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_dp_gpu.py tests/test_graph_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r2_t8.log 2>&1
rc=$?; echo "pytest rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r2_t8.log | tail -12 | cut -c1-300
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 700 python bench.py --no-cpu-baseline > gpurun_out/r2_b8.log 2> gpurun_out/r2_b8.err
echo "bench rc=$?"; tail -3 gpurun_out/r2_b8.err; python - <<PY
import json
l=[x for x in open('gpurun_out/r2_b8.log') if x.startswith('{')][-1]; d=json.loads(l)
print('ms', d['ms_per_step'], 'gemm', d['roofline']['gemm_ms_per_step'], d['roofline']['launches_per_step'], 'frac', d['roofline']['frac'], 'fusion', d['roofline']['fusion_mfma_util'], 'moe ms', d['moe_config'].get('ms_per_step'))
print('dp_model', d['dp_model'])
PY
