#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_dp_gpu.py -x -q > gpurun_out/s2_dp_tests2.log 2>&1 || { tail -40 gpurun_out/s2_dp_tests2.log; exit 1; }
tail -1 gpurun_out/s2_dp_tests2.log
for v in nowire wire nowire wire; do
  fl=""; [ $v = nowire ] && fl="--no-wire-optimizer"
  timeout -k 10 300 python bench.py --force-dist $fl --no-second-workload --no-cpu-baseline --no-roofline --steps 100 --warmup 20 > gpurun_out/s2_dpx_$v.log 2>&1 || { tail -5 gpurun_out/s2_dpx_$v.log; exit 1; }
  python - <<P
import json
l=json.loads(open('gpurun_out/s2_dpx_$v.log').read().strip().split('\n')[-1])
c=l['config']
print('$v', l['ms_per_step'], 'exposed', c.get('exposed_comm_ms'), c.get('segment_ms'))
P
done
