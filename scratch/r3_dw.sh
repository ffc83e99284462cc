#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -x -k "grouped_weight or k_rotation or fused_adamw or gemm" > $O/r3_dw_tests.log 2>&1; rc=$?
tail -5 $O/r3_dw_tests.log
if [ $rc -ne 0 ]; then grep -n "^E " $O/r3_dw_tests.log | head -20; fi
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python scratch/dw256_bench.py > $O/r3_dw_bench.log 2>&1; rc=$?; cat $O/r3_dw_bench.log | tail -5
if [ $rc -ge 124 ]; then exit $rc; fi
B="bench.py --no-cpu-baseline --no-second-workload --steps 60 --warmup 10"
for i in 1 2; do
  for v in 1 0; do
    VQA_DW256=$v timeout -k 10 300 python $B > $O/r3_dw_ab_${v}_$i.log 2>&1; rc=$?
    if [ $rc -ge 124 ]; then exit $rc; fi
    python - <<PY
import json
for l in open("$O/r3_dw_ab_${v}_$i.log"):
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print("dw256=$v run $i ms", d['ms_per_step'], 'gemm_ms', r['gemm_ms_per_step'], 'frac', r['frac'], 'fusion', r['fusion_mfma_util'])
PY
  done
done
