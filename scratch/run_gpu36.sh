#!/bin/bash
# tail runner: tests, then in-situ A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_blocks_gpu.py -x -q -k "tail or answer_head" -s > gpurun_out/t36a.log 2>&1 || { tail -30 gpurun_out/t36a.log; exit 1; }
tail -2 gpurun_out/t36a.log
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_graph_gpu.py tests/test_dp_gpu.py -x -q > gpurun_out/t36b.log 2>&1 || { tail -30 gpurun_out/t36b.log; exit 1; }
tail -2 gpurun_out/t36b.log
for rep in 1 2; do for tr in 1 0; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-roofline --tail-runner $tr > gpurun_out/r36.log 2> gpurun_out/r36.err || { tail -5 gpurun_out/r36.err; exit 1; }
  python - <<PY
import json
d=json.loads([x for x in open('gpurun_out/r36.log') if x.startswith('{')][-1])
print('tail=$tr rep $rep cfg2 ms', d['ms_per_step'], 'cfg3 ms', d['moe_config']['ms_per_step'])
PY
done; done
