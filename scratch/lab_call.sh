cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout -k 10 300 python scratch/norm_cov_tiny.py > $O/norm_cov_tiny.log 2>&1; grep -v amdgpu.ids $O/norm_cov_tiny.log | tail -8
timeout -k 10 900 python -m pytest tests/test_graph_gpu.py tests/test_dp_gpu.py tests/test_blocks_gpu.py -q -p no:cacheprovider > $O/sub_tests.log 2>&1; tail -4 $O/sub_tests.log
timeout -k 10 200 python bench.py --no-cpu-baseline --no-second-workload --no-roofline --workload cfg3_mcan_moe4 --steps 100 --warmup 15 2>/dev/null | python -c "import sys,json; [print('cfg3', json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]"
bash scratch/trace_gaps.sh
