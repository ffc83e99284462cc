#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in "--dtype fp16" "--eager --steps 20 --warmup 5" "--workload cfg1_concat" "--workload cfg3_mcan_moe4 --dtype fp16" "--torch-optimizer --steps 20 --warmup 5" "--batch 8 --steps 20 --warmup 5"; do
  timeout -k 10 400 python bench.py --no-cpu-baseline --no-second-workload --no-roofline $v > gpurun_out/r2_var.log 2> gpurun_out/r2_var.err
  echo "bench [$v] rc=$?"; grep "^{" gpurun_out/r2_var.log | tail -1 | python -c "import sys, json; d=json.loads(sys.stdin.read()); print('   ', d['ms_per_step'], 'ms', d['value'], 'samples/s', d['dtype'], d['config']['launch'][:60], d['config'].get('final_loss'), d['config'].get('loss_scale'))"; grep -iE "Traceback|Error" gpurun_out/r2_var.err | head -3
done
