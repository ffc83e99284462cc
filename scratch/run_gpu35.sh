#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for rep in 1 2; do for lib in "" scratch/libvqa_nt0.so; do
  if [ -n "$lib" ]; then export VQA_HIP_LIB=$GRAFT_REPO_ROOT/$lib; else unset VQA_HIP_LIB; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r2_nt.log 2> gpurun_out/r2_nt.err || exit 1
  python - <<PY
import json
d=json.loads([x for x in open('gpurun_out/r2_nt.log') if x.startswith('{')][-1])
print('lib=[$lib] rep $rep cfg2 ms', d['ms_per_step'], 'cfg3 ms', d['moe_config']['ms_per_step'])
PY
done; done
