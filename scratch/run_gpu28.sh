#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_dp_gpu.py -m gpu -q -p no:cacheprovider > gpurun_out/r2_t28.log 2>&1
rc=$?; echo "dp tests rc=$rc"; grep -E "passed|failed|^FAILED|^ERROR|^E  " gpurun_out/r2_t28.log | tail -10 | cut -c1-300
if [ $rc -ne 0 ]; then exit $rc; fi
export VQA_BENCH_REHEARSE=1
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-second-workload --no-roofline > gpurun_out/r2_reh.log 2> gpurun_out/r2_reh.err
echo "rehearsal rc=$?"; grep "^{" gpurun_out/r2_reh.log | tail -1 | python -c "import sys, json; d=json.loads(sys.stdin.read()); print(d['config']['segment_bytes'], d['config']['allreduce_bytes'], d['config']['final_loss'])"
unset VQA_BENCH_REHEARSE
timeout -k 10 400 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/r2_b28.log 2> gpurun_out/r2_b28.err || exit 1
python - <<PY
import json
d=json.loads([x for x in open('gpurun_out/r2_b28.log') if x.startswith('{')][-1])
print('cfg2', d['ms_per_step'], 'cfg3', d['moe_config']['ms_per_step']); print(json.dumps(d['dp_model'])[:900])
PY
