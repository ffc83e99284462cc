#!/bin/bash
# same-box A/B of the cfg2 step between the default library and scratch/libvqa_$1.so (interleaved A B A B), then optional extra commands
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-second-workload --no-roofline --steps 80 --warmup 15"
for i in 1 2; do
  for v in default $1; do
    if [ $v = default ]; then unset VQA_HIP_LIB; else export VQA_HIP_LIB=$R/scratch/libvqa_$v.so; fi
    timeout -k 10 200 $B 2>/dev/null | python -c "import sys,json; [print('$v', json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]" || exit 1
  done
done
unset VQA_HIP_LIB
