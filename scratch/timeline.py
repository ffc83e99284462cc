"""Analyse a rocprofv3 kernel trace (sqlite) of bench.py: per-step busy time, overlap, per-kernel stats in the LAST steps."""
import sqlite3, sys, collections
db = sys.argv[1]; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute(f"select s.kernel_name, d.start, d.end, d.queue_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start").fetchall()
# steps are delimited by the adamw kernel (last kernel of a step): take the last nsteps complete steps
ends = [i for i, r in enumerate(rows) if 'adamw_multi' in r[0]]
ends = ends[1::2] if len(ends) >= 2 and rows[ends[0]][0] == rows[ends[1]][0] else ends      # two groups -> two launches per step
sel = rows[ends[-nsteps - 1] + 1: ends[-1] + 1]
t0, t1 = sel[0][1], sel[-1][2]
print(f'{nsteps} steps: wall {(t1 - t0) / 1e6 / nsteps:.3f} ms/step, {len(sel) / nsteps:.0f} kernels/step, queues {sorted({r[3] for r in sel})}')
# union busy and sum
ev = sorted([(r[1], 1) for r in sel] + [(r[2], -1) for r in sel])
busy = 0; depth = 0; last = None; hist = collections.Counter()
for t, d in ev:
    if depth > 0: busy += t - last; hist[min(depth, 4)] += t - last
    depth += d; last = t
tot = sum(r[2] - r[1] for r in sel)
print(f'sum of kernel durations {tot / 1e6 / nsteps:.3f} ms/step, union busy {busy / 1e6 / nsteps:.3f} ms/step, idle {(t1 - t0 - busy) / 1e6 / nsteps:.3f} ms/step')
print('time at concurrency 1/2/3/4+ (ms/step):', [round(hist[k] / 1e6 / nsteps, 3) for k in (1, 2, 3, 4)])
agg = collections.defaultdict(lambda: [0, 0])
for r in sel:
    agg[r[0]][0] += 1; agg[r[0]][1] += r[2] - r[1]
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f'{t / tot * 100:5.1f}%  n/step={n / nsteps:6.1f} avg={t / n / 1e3:7.1f}us  {k[:110]}')
