"""HBM-side bytes per GEMM launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: 3 + 2 of the 4
TCC slots, MI355X_MICROARCH.md "Counter slots") of the SAME bench.py command; writes profiles/<round>/gemm_traffic.json.

    python profiles/pmc_traffic.py <fetch_dir> <write_dir> <out.json> <steps_profiled> "<command>"

Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE tallies 128-B requests at 64 B -> doubled; WRITE_SIZE is read as
is (16-B-per-lane streaming stores and float atomics are exact).  Counter unit: KiB.  Validated on a kernel of known traffic in the
same run: adamw_multi_kernel moves 16 B read + 14 B written per parameter.
"""
import csv
import glob
import json
import os
import sys

GEMM = ('gemm_v1_kernel', 'gemm_v1_grouped_kernel', 'gemm_kernel', 'gemm_ws_kernel', 'fused_inproj_attn_kernel')


def collect(d, counter):
    tot, n, adam = 0.0, 0, 0.0
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit(f'no counter_collection.csv under {d}')
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') != counter:
                continue
            name = row.get('Kernel_Name', '')
            v = float(row['Counter_Value'])
            if any(g in name for g in GEMM):
                tot += v
                n += 1
            elif 'adamw_multi_kernel' in name:
                adam += v
    return tot, n, adam


def main():
    fetch_dir, write_dir, out, steps, cmd = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
    f_kb, n_f, f_adam = collect(fetch_dir, 'FETCH_SIZE')
    w_kb, n_w, w_adam = collect(write_dir, 'WRITE_SIZE')
    assert n_f == n_w and n_f > 0, (n_f, n_w)
    bytes_total = (2.0 * f_kb + w_kb) * 1024.0
    res = {
        'command': cmd,
        'steps_profiled': steps,
        'gemm_launches': n_f,
        'fetch_size_kb_sum': f_kb, 'write_size_kb_sum': w_kb,
        'correction': 'FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); WRITE_SIZE as read',
        'hbm_bytes_per_gemm_launch': int(bytes_total / n_f),
        'hbm_gb_per_step_gemm': round(bytes_total / steps / 1e9, 2),
        'adamw_check': {'fetch_gb_per_step_corrected': round(2.0 * f_adam * 1024 / steps / 1e9, 2), 'write_gb_per_step': round(w_adam * 1024 / steps / 1e9, 2),
                        'expected': '16 B read + 14 B written per parameter (p, g, m, v in; p, m, v + bf16 shadow out)'},
    }
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
