#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in off 4 8 12 16 24 48 off; do
    fl="--prefetch-wgs $v"; [ $v = off ] && fl="--no-weight-prefetch"
    timeout -k 10 300 python bench.py $fl --no-second-workload --no-cpu-baseline --no-roofline > gpurun_out/s2_pf_$v.log 2>&1 || { tail -5 gpurun_out/s2_pf_$v.log; exit 1; }
    python - <<P
import json
l=json.loads(open('gpurun_out/s2_pf_$v.log').read().strip().split('\n')[-1])
print('prefetch $v', l['ms_per_step'])
P
done
