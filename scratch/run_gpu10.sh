#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "gemm" > gpurun_out/r2_t10.log 2>&1
rc=$?; echo "pytest rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r2_t10.log | tail -8 | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python scratch/skinny_bench.py > gpurun_out/r2_skinny.log 2>&1; cat gpurun_out/r2_skinny.log
for i in 1 2; do for sk in 0 1; do
  timeout -k 10 300 python bench.py --workload cfg3_mcan_moe4 --no-cpu-baseline --no-second-workload --no-roofline --gemm-skinny $sk > gpurun_out/r2_sk_$sk$i.log 2> gpurun_out/r2_sk_$sk$i.err || exit 1
  python - <<PY
import json
l=[x for x in open('gpurun_out/r2_sk_$sk$i.log') if x.startswith('{')][-1]; d=json.loads(l)
print('skinny=$sk run $i cfg3 ms', d['ms_per_step'])
PY
done; done
