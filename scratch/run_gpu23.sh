#!/bin/bash
# rehearsal of bench.py's N > 1 path on one GPU (2 ranks on cuda:0 over gloo): default flags, then fp32 buckets, then the MoE config
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export VQA_BENCH_REHEARSE=1
for v in "" "--grad-dtype fp32" "--workload cfg3_mcan_moe4"; do
  timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-second-workload --no-roofline $v > gpurun_out/r2_reh.log 2> gpurun_out/r2_reh.err
  echo "rehearsal [$v] rc=$?"; grep "^{" gpurun_out/r2_reh.log | tail -1 | cut -c1-1500; grep -iE "error|Traceback|capture failed" gpurun_out/r2_reh.err | head -5
done
