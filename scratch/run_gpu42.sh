#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 400 python scratch/gen_bench.py 32 > gpurun_out/gen42.log 2>&1 || { tail -20 gpurun_out/gen42.log; exit 1; }
grep generative gpurun_out/gen42.log
timeout -k 10 700 python -m pytest tests/test_generative_gpu.py tests/test_graph_gpu.py -x -q > gpurun_out/t42.log 2>&1 || { tail -40 gpurun_out/t42.log; exit 1; }
tail -2 gpurun_out/t42.log
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/b42.log 2> gpurun_out/b42.err || { tail -20 gpurun_out/b42.err; exit 1; }
python - <<PY
import json
d=json.loads([x for x in open('gpurun_out/b42.log') if x.startswith('{')][-1])
print('cfg2', d['ms_per_step'], 'frac', d['roofline']['frac'], 'cfg3', d['moe_config']['ms_per_step'])
print('generative', json.dumps(d['generative_config']))
PY
