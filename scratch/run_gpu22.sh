#!/bin/bash
# end-of-round evidence at HEAD: full GPU suite, profiles, default bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r2_t22.log 2>&1
rc=$?; echo "all gpu tests rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r2_t22.log | tail -8 | cut -c1-300
bash profiles/collect.sh r02 > gpurun_out/r2_collect.log 2>&1 || echo "collect failed"
tail -3 gpurun_out/r2_collect.log | cut -c1-300
timeout -k 10 500 python bench.py > gpurun_out/r2_bfinal.log 2> gpurun_out/r2_bfinal.err || exit 1
tail -1 gpurun_out/r2_bfinal.log | cut -c1-600
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
