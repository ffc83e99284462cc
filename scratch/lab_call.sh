cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout -k 10 200 python scratch/epi_diff.py > $O/epi_diff.log 2>&1; tail -12 $O/epi_diff.log
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -p no:cacheprovider -k "gemm or compact or grouped or specialised" > $O/fast_tests.log 2>&1; tail -3 $O/fast_tests.log
