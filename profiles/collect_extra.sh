#!/bin/bash
# Round-3 extras beside profiles/collect.sh (run through gpurun from the repo root AFTER collect.sh):   profiles/collect_extra.sh r03
#   one-rank RCCL logs of the segmented data-parallel step (cfg2 and cfg3), the default bench line with the CPU baseline, per-(kernel, grid)
#   launch statistics, SQ_WAIT_ANY / SQ_WAVE_CYCLES of the GEMM kernels, the 256 x 256 weight-gradient microbenchmark, generative kernel stats,
#   the strict north-star table the parity tests print.
R=${GRAFT_REPO_ROOT:-/root/repo}
rnd=${1:-r03}
out=$R/gpurun_out/prof_$rnd
mkdir -p $out
cd $R
B="python bench.py --no-cpu-baseline --no-second-workload --no-roofline --steps 100 --warmup 20"
timeout -k 10 300 $B --force-dist > $out/rccl_one_rank_cfg2.log 2> $out/rccl_one_rank_cfg2.err || exit 1
timeout -k 10 300 $B --force-dist --workload cfg3_mcan_moe4 > $out/rccl_one_rank_cfg3.log 2> $out/rccl_one_rank_cfg3.err || exit 1
timeout -k 10 300 $B --force-dist --grad-dtype fp32 > $out/rccl_one_rank_cfg2_fp32.log 2>/dev/null || exit 1
timeout -k 10 600 python bench.py > $out/final_bench_default.log 2> $out/final_bench_default.err || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --no-second-workload --dtype fp16 > $out/bench_fp16.log 2>/dev/null || exit 1
timeout -k 10 200 python scratch/dw256_bench.py > $out/dw256_bench.log 2>&1 || exit 1
bash scratch/trace_by_shape.sh > /dev/null 2>&1 && cp gpurun_out/trace_shape/by_shape.txt $out/eager_launches_by_shape.txt
bash scratch/pmc_wait.sh > /dev/null 2>&1 && cp gpurun_out/pmc_wait/wait_ratio.txt $out/pmc_wait_ratio.txt
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/gen --output-format csv -- python3 $R/scratch/gen_bench.py > $out/generative.log 2>&1) && { f=$(find $out/gen -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/generative_kernel_stats.csv; }
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/graph_only --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-second-workload --no-roofline --steps 300 --warmup 5 > $out/graph_only.log 2>&1) && { f=$(find $out/graph_only -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/graph_only_kernel_stats.csv; }
rm -rf $out/graph_only
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -q -s -p no:cacheprovider  > $out/parity_run.log 2>&1
grep -E "^(PARITY|ROTATION|NORTH-STAR)|passed|failed" $out/parity_run.log > $out/north_star_table.txt
rm -rf $out/gen $out/graph $out/eager $out/cfg3 $out/pmc_fetch $out/pmc_write $out/pmc_mfma      # raw traces: tens of MB
ls $out
