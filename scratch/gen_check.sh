#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -q -p no:cacheprovider -k "patchify or clip or expert_row" > $O/gen_check.log 2>&1; tail -3 $O/gen_check.log
for i in 1 2; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-second-workload --no-roofline --steps 100 --warmup 15 2>/dev/null | python -c "import sys,json; [print('cfg2', json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"; done
