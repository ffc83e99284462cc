#!/bin/bash
# SQ_WAIT_ANY / SQ_WAVE_CYCLES of the GEMM kernels in the eager cfg2 step (own PMC pass, kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/pmc_wait
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $out/run --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-second-workload --eager --no-roofline --steps 3 --warmup 2 > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
python3 - <<PY
import csv, glob, collections, re
f = glob.glob("$out/run/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'gemm_v1' not in n and 'fused_inproj' not in n and 'adamw_multi' not in n: continue
    n = re.sub(r'\(anonymous namespace\)::', '', n); n = re.sub(r'\(.*', '', n)[:64]
    agg[n][r['Counter_Name']] += float(r['Counter_Value'])
with open("$out/wait_ratio.txt", 'w') as fo:
    for n, c in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
        if c.get('SQ_WAVE_CYCLES'):
            line = '%-66s SQ_WAIT_ANY / SQ_WAVE_CYCLES = %.3f' % (n, c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES'])
            print(line); fo.write(line + '\n')
PY
