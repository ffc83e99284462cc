#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider -x -k "grouped_weight" 2>&1 | tail -3
timeout -k 10 300 python scratch/dw256_bench.py 2>&1 | tail -2 | sed 's/^/prefetch: /'
VQA_HIP_LIB=$R/scratch/libvqa_dwnopf.so timeout -k 10 300 python scratch/dw256_bench.py 2>&1 | tail -2 | sed 's/^/no prefetch: /'
VQA_HIP_LIB=$R/scratch/libvqa_dwtrace.so DW_MODE=mix timeout -k 10 200 python scratch/dw_trace.py 2>/dev/null | sed 's/^/first: /' | tail -6
VQA_HIP_LIB=$R/scratch/libvqa_dwtracelast.so DW_MODE=mix timeout -k 10 200 python scratch/dw_trace.py 2>/dev/null | sed 's/^/last: /' | tail -6
B="bench.py --no-cpu-baseline --no-second-workload --steps 60 --warmup 10"
for i in 1 2; do
  for v in 1 0; do
    VQA_DW256=$v timeout -k 10 300 python $B > $O/r3_dw7_ab_${v}_$i.log 2>&1; rc=$?
    if [ $rc -ge 124 ]; then exit $rc; fi
    python - <<PY
import json
for l in open("$O/r3_dw7_ab_${v}_$i.log"):
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; print("dw256=$v run $i ms", d['ms_per_step'], 'gemm_ms', r['gemm_ms_per_step'], 'frac', r['frac'], 'fusion', r['fusion_mfma_util'])
PY
  done
done
