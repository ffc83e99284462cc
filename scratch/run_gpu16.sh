#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r2_t16.log 2>&1
rc=$?; echo "all gpu tests rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r2_t16.log | tail -12 | cut -c1-300
timeout -k 10 400 python bench.py > gpurun_out/r2_b16.log 2> gpurun_out/r2_b16.err || exit 1
python - <<PY
import json
l=[x for x in open('gpurun_out/r2_b16.log') if x.startswith('{')][-1]; d=json.loads(l)
print('cfg2 ms', d['ms_per_step'], 'value', d['value'], 'roofline', {k: d['roofline'].get(k) for k in ('frac','gemm_ms_per_step','fusion_mfma_util','traffic')})
print('cfg3', (d.get('moe_config') or {}).get('ms_per_step'))
print('dp_model', json.dumps(d.get('dp_model'))[:900])
print('cpu', d.get('cpu_baseline'))
PY
