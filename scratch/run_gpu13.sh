#!/bin/bash
# fused-attention DMA fix (exactness + microbench), MoE branch modes A/B, graph / dp tests with branches
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "fused_inproj" > gpurun_out/r2_t13a.log 2>&1
rc=$?; echo "fused kernel tests rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r2_t13a.log | tail -8 | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 900 python -m pytest tests/test_graph_gpu.py tests/test_dp_gpu.py tests/test_parity_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r2_t13b.log 2>&1
rc=$?; echo "graph/dp/parity tests rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r2_t13b.log | tail -8 | cut -c1-300
timeout -k 10 300 python scratch/dbg_moe_graph.py > gpurun_out/r2_dbg_moe_graph.log 2>&1; grep -E "eager|graph" gpurun_out/r2_dbg_moe_graph.log | cut -c1-200
timeout -k 10 300 python scratch/fused_attn_bench.py > gpurun_out/r2_fused_attn_bench.log 2>&1; cat gpurun_out/r2_fused_attn_bench.log
for i in 1 2; do for br in 0 1 2; do
  timeout -k 10 300 python bench.py --workload cfg3_mcan_moe4 --no-cpu-baseline --no-second-workload --no-roofline --moe-branches $br > gpurun_out/r2_br_$br$i.log 2> gpurun_out/r2_br_$br$i.err || exit 1
  python - <<PY
import json
l=[x for x in open('gpurun_out/r2_br_$br$i.log') if x.startswith('{')][-1]; d=json.loads(l)
print('moe_branches=$br run $i cfg3 ms', d['ms_per_step'])
PY
done; done
