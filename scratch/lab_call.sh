cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -q -x -p no:cacheprovider -k "gemm or compact or grouped or specialised" > $O/fast_tests.log 2>&1; tail -3 $O/fast_tests.log
for i in 1 2; do for f in 1 5; do VQA_GEMM_FAST=$f timeout -k 10 200 python bench.py --no-cpu-baseline --no-second-workload --no-roofline --steps 80 --warmup 15 2>/dev/null | python -c "import sys,json; [print('fast=$f', json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]" || exit 1; done; done
timeout -k 10 60 python scratch/lab_run.py scratch/lab_64x64.so NN 2048 768 768 > $O/lab_epi2.log 2>&1
grep "lab_\|epi\|wg total" $O/lab_epi2.log
