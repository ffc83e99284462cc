#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_dp_gpu.py tests/test_graph_gpu.py -m gpu -q -p no:cacheprovider -s > gpurun_out/r2_t18.log 2>&1
rc=$?; echo "dp/graph tests rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  |^[A-E] " gpurun_out/r2_t18.log | tail -16 | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/r2_b18.log 2> gpurun_out/r2_b18.err || exit 1
python - <<PY
import json
l=[x for x in open('gpurun_out/r2_b18.log') if x.startswith('{')][-1]; d=json.loads(l)
print('cfg2 ms', d['ms_per_step'], 'cfg3', (d.get('moe_config') or {}).get('ms_per_step'))
print('dp_model', json.dumps(d.get('dp_model'))[:1200])
PY
