#!/bin/bash
# round 3, first GPU call: suite at the round's first edits, default bench, runtime env-knob sweep
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/r3_tests1.log 2>&1; rc=$?
tail -5 $O/r3_tests1.log
if [ $rc -ge 124 ]; then echo "pytest killed rc=$rc"; exit $rc; fi
B="bench.py --no-cpu-baseline --no-second-workload --no-roofline --steps 60 --warmup 10"
run() { name=$1; shift; echo "== $name"; env "$@" timeout -k 10 240 python $B > $O/r3_env_$name.log 2>&1; rc=$?; if [ $rc -ge 124 ]; then echo "killed rc=$rc"; exit $rc; fi; python - <<PY
import json
for l in open("$O/r3_env_$name.log"):
    if l.startswith('{'):
        d = json.loads(l); print("$name", d['ms_per_step'], d['value'])
PY
}
run base A=1
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq2 GPU_MAX_HW_QUEUES=2
run pktcap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run base2 A=1
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/r3_bench_default1.log 2> $O/r3_bench_default1.err; tail -c 3000 $O/r3_bench_default1.log
