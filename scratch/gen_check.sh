#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_graph_gpu.py -q -x -p no:cacheprovider -k "adamw or graph or replayed or checkpoint or norm" > $O/gen_check.log 2>&1; tail -4 $O/gen_check.log
for i in 1 2; do for v in 1 0; do VQA_SPARSE_ROWS=$v timeout -k 10 200 python bench.py --no-cpu-baseline --no-second-workload --no-roofline --steps 100 --warmup 15 2>/dev/null | python -c "import sys,json; [print('sparse_rows=$v cfg2', json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"; done; done
