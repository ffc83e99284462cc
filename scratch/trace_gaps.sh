#!/bin/bash
# gaps between consecutive kernels of one queue inside the captured cfg2 step (rocprofv3 --kernel-trace of graph replays)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/trace_gaps
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $out/run --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-second-workload --no-roofline --steps 40 --warmup 5 > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
python3 - <<PY
import csv, glob, collections, re
import statistics as st
f = glob.glob("$out/run/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print('columns', list(rows[0].keys()))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 30 % of the trace = steady-state graph replays
t0 = int(rows[int(len(rows) * 0.6)]['Start_Timestamp'])
rows = [r for r in rows if int(r['Start_Timestamp']) >= t0]
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get('Queue_Id', '0')].append(r)
span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e6
busy_any = 0
# union of busy intervals over all queues
iv = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows)
cur_s, cur_e = iv[0]
for s, e in iv[1:]:
    if s > cur_e:
        busy_any += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy_any += cur_e - cur_s
print(f'window {span:.2f} ms, kernels {len(rows)}, GPU busy with at least one kernel {busy_any / 1e6:.2f} ms = {busy_any / 1e6 / span * 100:.1f} %')
with open("$out/gaps.txt", 'w') as fo:
    for q, rs in byq.items():
        gaps = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(rs, rs[1:])]
        gaps = [g for g in gaps if g < 200000]
        dur = [int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs]
        if not gaps: continue
        line = f'queue {q}: kernels {len(rs)}, kernel time {sum(dur) / 1e6:.2f} ms, gaps: median {st.median(gaps) / 1e3:.2f} us, mean {sum(gaps) / len(gaps) / 1e3:.2f} us, sum {sum(gaps) / 1e6:.2f} ms, negative (overlap) {sum(g < 0 for g in gaps)}'
        print(line); fo.write(line + '\n')
        hist = collections.Counter(min(int(g // 1000), 20) for g in gaps)
        print('   gap histogram (us: count)', sorted(hist.items()))
PY
