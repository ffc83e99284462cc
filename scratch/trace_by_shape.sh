#!/bin/bash
# per-(kernel, grid) launch statistics of the eager cfg2 step (rocprofv3 --kernel-trace), to compare in-situ launches with the microbenchmarks
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/trace_shape
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/eager --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-second-workload --eager --no-roofline --steps 6 --warmup 2 > $out/eager.log 2>&1 || { tail -5 $out/eager.log; exit 1; }
python3 - <<PY
import csv, glob, collections, re
f = glob.glob("$out/eager/**/*kernel_trace.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'\(.*', '', n)[:70]
    agg[(n, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size',''), r.get('Workgroup_Size_X', r.get('Workgroup_Size','')))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
steps = 8
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
tot = sum(sum(v) for v in agg.values())
print('total kernel ms/step %.3f' % (tot / 1e6 / steps))
with open("$out/by_shape.txt", 'w') as fo:
    for (n, g, w), v in rows[:60]:
        line = '%-72s grid %8s wg %4s  n/step %6.1f  avg %7.1f us  ms/step %6.3f' % (n, g, w, len(v) / steps, sum(v) / len(v) / 1e3, sum(v) / 1e6 / steps)
        print(line); fo.write(line + '\n')
PY
