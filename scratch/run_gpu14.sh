#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_graph_gpu.py tests/test_dp_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r2_t14a.log 2>&1
rc=$?; echo "graph/dp tests rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r2_t14a.log | tail -12 | cut -c1-300
timeout -k 10 300 python scratch/dbg_moe_graph.py > gpurun_out/r2_dbg_moe_graph.log 2>&1; grep -E "eager|graph" gpurun_out/r2_dbg_moe_graph.log | cut -c1-200
bash profiles/collect.sh r02 || echo "collect failed"
ls gpurun_out/prof_r02 | head -30
