#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_dp_gpu.py -x -q > gpurun_out/s2_dp_tests.log 2>&1 || { tail -40 gpurun_out/s2_dp_tests.log; exit 1; }
tail -1 gpurun_out/s2_dp_tests.log
for v in towers depth towers depth; do
  timeout -k 10 300 python bench.py --force-dist --dp-split $v --no-second-workload --no-cpu-baseline --no-roofline --steps 100 --warmup 20 > gpurun_out/s2_dp_$v.log 2>&1 || { tail -5 gpurun_out/s2_dp_$v.log; exit 1; }
  python - <<P
import json
l=json.loads(open('gpurun_out/s2_dp_$v.log').read().strip().split('\n')[-1])
c=l['config']
print('$v', l['ms_per_step'], 'exposed', c.get('exposed_comm_ms'), c.get('segment_ms'), {k: round(v/1e6,1) for k,v in c.get('segment_bytes',{}).items()})
P
done
timeout -k 10 300 python bench.py --no-second-workload --no-cpu-baseline --no-roofline > gpurun_out/s2_dp_single.log 2>&1; python -c "
import json; l=json.loads(open('gpurun_out/s2_dp_single.log').read().strip().split('\n')[-1]); print('single graph', l['ms_per_step'])"
