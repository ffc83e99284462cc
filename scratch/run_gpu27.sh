#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_data_feed_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r2_t27.log 2>&1
rc=$?; echo "feed tests rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r2_t27.log | tail -6 | cut -c1-300
timeout -k 10 400 python scratch/feed_bench.py > gpurun_out/r2_feed.log 2>&1; grep -E "cfg2|Error|error" gpurun_out/r2_feed.log | tail -4 | cut -c1-400
