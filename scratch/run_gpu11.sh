#!/bin/bash
# fused in-projection + attention: exactness, block / parity tests, in-call A/B of the training step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "fused_inproj or attention" > gpurun_out/r2_t11a.log 2>&1
rc=$?; echo "kernel tests rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r2_t11a.log | tail -8 | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 python -m pytest tests/test_blocks_gpu.py tests/test_parity_gpu.py tests/test_dp_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r2_t11b.log 2>&1
rc=$?; echo "block/parity/dp tests rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r2_t11b.log | tail -8 | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
for i in 1 2; do for fa in 0 1 3; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-second-workload --fused-attn $fa > gpurun_out/r2_fa_$fa$i.log 2> gpurun_out/r2_fa_$fa$i.err || exit 1
  python - <<PY
import json
l=[x for x in open('gpurun_out/r2_fa_$fa$i.log') if x.startswith('{')][-1]; d=json.loads(l)
r=d.get('roofline',{})
print('fused_attn=$fa run $i cfg2 ms', d['ms_per_step'], 'gemm_ms', r.get('gemm_ms_per_step'), 'frac', r.get('frac'), 'fusion_util', r.get('fusion_mfma_util'))
PY
done; done
