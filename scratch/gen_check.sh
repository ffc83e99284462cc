#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_blocks_gpu.py -q -x -p no:cacheprovider -k "layernorm or clip_layer or roberta_layer or cross_modal" > $O/gen_check.log 2>&1; tail -3 $O/gen_check.log
for i in 1 2; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-second-workload --no-roofline --steps 100 --warmup 15 2>/dev/null | python -c "import sys,json; [print('cfg2', json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"; done
cd /tmp && export TMPDIR=/tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/lnr --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-second-workload --no-roofline --steps 100 --warmup 5 > $O/lnr.log 2>&1; f=$(find $O/lnr -name '*kernel_stats.csv' | head -1); grep "ln_bwd_reduce\|clip_assemble\|rows_mask" $f | cut -c1-160; rm -rf $O/lnr
