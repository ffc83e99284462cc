#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -s --maxfail=10 -p no:cacheprovider > gpurun_out/r2_t7.log 2>&1
rc=$?
echo "pytest rc=$rc" >> gpurun_out/r2_t7.log
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r2_t7.log | tail -12
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/r2_b7.log 2> gpurun_out/r2_b7.err
echo "bench rc=$?"; python - <<PY
import json
l=[x for x in open('gpurun_out/r2_b7.log') if x.startswith('{')][-1]; d=json.loads(l)
print('ms', d['ms_per_step'], 'gemm', d['roofline']['gemm_ms_per_step'], d['roofline']['launches_per_step'], 'frac', d['roofline']['frac'], 'fusion', d['roofline']['fusion_mfma_util'], 'moe ms', d['moe_config']['ms_per_step'])
PY
