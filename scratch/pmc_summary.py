"""Summarise scratch/pmc_gemm.sh output: per configuration, the mean of every counter over the gemm launches."""
import csv, glob, os, sys, collections
root = sys.argv[1]
rows = collections.OrderedDict()
for d in sorted(glob.glob(os.path.join(root, '*_*'))):
    if not os.path.isdir(d): continue
    tag = os.path.basename(d).rsplit('_', 1)[0]
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'gemm' not in r['Kernel_Name']: continue
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
        for k, v in acc.items():
            v = v[len(v) // 4:]              # skip the cold launches
            rows.setdefault(tag, {})[k] = sum(v) / len(v)
    for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
        t = [float(r['End_Timestamp']) - float(r['Start_Timestamp']) for r in csv.DictReader(open(f)) if 'gemm' in r['Kernel_Name']]
        t = t[len(t) // 4:]
        rows.setdefault(tag, {}).setdefault('us', []).append(sum(t) / len(t) / 1e3)
for tag, c in rows.items():
    us = sum(c.pop('us')) / 3 if 'us' in c else 0
    print(f'== {tag}   {us:.1f} us (under counters)')
    for k in sorted(c): print(f'   {k:34s} {c[k]:16.0f}')
    if 'TCC_HIT_sum' in c: print(f"   L2 hit rate {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}   EA read MB (x128B) {c['TCC_EA0_RDREQ_sum'] * 128 / 1e6:.1f}   TCP->TCC read MB (x128) {c.get('TCP_TCC_READ_REQ_sum', 0) * 128 / 1e6:.1f}")
    if 'SQ_WAVE_CYCLES' in c: print(f"   wait_any {c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.2f}  wait_inst {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES']:.2f}  active {c['SQ_ACTIVE_INST_ANY'] / c['SQ_WAVE_CYCLES']:.2f}")
