#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -p no:cacheprovider -k "grouped" > gpurun_out/r2_t21.log 2>&1; grep -E "passed|failed" gpurun_out/r2_t21.log | tail -2
for i in 1 2; do for gp in 0 512 768 1024; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --group-persistent $gp > gpurun_out/r2_gp_$gp$i.log 2> gpurun_out/r2_gp_$gp$i.err || exit 1
  python - <<PY
import json
l=[x for x in open('gpurun_out/r2_gp_$gp$i.log') if x.startswith('{')][-1]; d=json.loads(l)
print('group_persistent=$gp run $i cfg2 ms', d['ms_per_step'], 'cfg3 ms', (d.get('moe_config') or {}).get('ms_per_step'))
PY
done; done
