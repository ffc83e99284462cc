#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=25 -s > $O/r3_tests_${1:-6}.log 2>&1; rc=$?
grep -E "passed|failed|FAILED|ERROR" $O/r3_tests_${1:-6}.log | tail -30
if [ $rc -ge 124 ]; then echo "pytest killed rc=$rc"; exit $rc; fi
timeout -k 10 200 python __graft_entry__.py smoke 2>&1 | tail -2
timeout -k 10 500 python bench.py --no-cpu-baseline > $O/r3_bench_${1:-6}.log 2> $O/r3_bench_${1:-6}.err; python - <<PY
import json
for l in open("$O/r3_bench_${1:-6}.log"):
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('cfg2 ms', d['ms_per_step'], 'value', d['value'], 'gemm_ms', r['gemm_ms_per_step'], 'frac', r['frac'], 'fusion', r['fusion_mfma_util'])
        print('cfg3', d['moe_config'].get('ms_per_step'), d['moe_config'].get('launch'), 'gen', d['generative_config'].get('ms_per_step'))
        print('dp_model', json.dumps(d['dp_model'])[:600])
PY
grep -i "empty\|error" $O/r3_bench_${1:-6}.err | head -5
