#!/bin/bash
# expert runners + dense combine: exactness, then cfg3 A/B in one call; fused-attention microbench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_blocks_gpu.py -m gpu -q -x -p no:cacheprovider -k "moe" -s > gpurun_out/r2_t12a.log 2>&1
rc=$?; echo "expert tests rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  |expert runner|^moe " gpurun_out/r2_t12a.log | tail -30 | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 python -m pytest tests/test_graph_gpu.py tests/test_parity_gpu.py tests/test_dp_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r2_t12b.log 2>&1
rc=$?; echo "graph/parity/dp tests rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r2_t12b.log | tail -8 | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 python scratch/fused_attn_bench.py > gpurun_out/r2_fused_attn_bench.log 2>&1; cat gpurun_out/r2_fused_attn_bench.log
for i in 1 2; do for er in 0 1 2; do
  timeout -k 10 300 python bench.py --workload cfg3_mcan_moe4 --no-cpu-baseline --no-second-workload --no-roofline --expert-runners $((er>0)) --moe-branches $((er>1)) > gpurun_out/r2_er_$er$i.log 2> gpurun_out/r2_er_$er$i.err || exit 1
  python - <<PY
import json
l=[x for x in open('gpurun_out/r2_er_$er$i.log') if x.startswith('{')][-1]; d=json.loads(l)
print('expert_runners=$er run $i cfg3 ms', d['ms_per_step'])
PY
done; done
