#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python scratch/gemm_mid_m_bench.py > gpurun_out/r2_mid_m.log 2>&1; cat gpurun_out/r2_mid_m.log
timeout -k 10 600 python -m pytest tests/test_graph_gpu.py tests/test_dp_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r2_t15a.log 2>&1
rc=$?; echo "graph/dp tests rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r2_t15a.log | tail -12 | cut -c1-300
