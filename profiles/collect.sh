#!/bin/bash
# Collects the round's profiler evidence on a GPU box (run through gpurun from the repo root):
#   profiles/collect.sh r02
# 1. rocprofv3 --kernel-trace --stats of the default bench.py command (captured graph) and of the eager step, cfg2 and cfg3
# 2. PMC: FETCH_SIZE and WRITE_SIZE in their own passes (kernel-trace only) on the eager step -> gemm_traffic.json
# Everything lands in gpurun_out/prof_<round>/; copy the summaries into profiles/<round>/ afterwards (cp lines at the end do it).
R=${GRAFT_REPO_ROOT:-/root/repo}
rnd=${1:-r02}
out=$R/gpurun_out/prof_$rnd
mkdir -p $out $R/profiles/$rnd
cd /tmp && export TMPDIR=/tmp
BENCH="$R/bench.py --no-cpu-baseline --no-second-workload"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/graph --output-format csv -- python3 $BENCH --steps 30 --warmup 5 > $out/graph.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/eager --output-format csv -- python3 $BENCH --eager --no-roofline --steps 10 --warmup 3 > $out/eager.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $out/cfg3 --output-format csv -- python3 $BENCH --workload cfg3_mcan_moe4 --no-roofline --steps 30 --warmup 5 > $out/cfg3.log 2>&1 || exit 1
PMC_CMD="$BENCH --eager --no-roofline --steps 3 --warmup 2"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch --output-format csv -- python3 $PMC_CMD > $out/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write --output-format csv -- python3 $PMC_CMD > $out/pmc_write.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d $out/pmc_mfma --output-format csv -- python3 $PMC_CMD > $out/pmc_mfma.log 2>&1 || echo "mfma counter pass failed (non-fatal)"
python3 $R/profiles/pmc_traffic.py $out/pmc_fetch $out/pmc_write $R/profiles/$rnd/gemm_traffic.json 5 "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --no-cpu-baseline --no-second-workload --eager --no-roofline --steps 3 --warmup 2 (two separate passes)" || exit 1
cp $R/profiles/$rnd/gemm_traffic.json $out/gemm_traffic.json
for k in graph eager cfg3; do f=$(find $out/$k -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $out/${k}_kernel_stats.csv; done
# raw PMC rows of the GEMM kernels only (the full per-dispatch CSVs are tens of MB)
python3 - <<PY
import csv, glob, os
out = "$out"
for tag in ('pmc_fetch', 'pmc_write', 'pmc_mfma'):
    rows = []
    for f in glob.glob(os.path.join(out, tag, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r.get('Kernel_Name', '')
            if 'gemm' in n or 'fused_inproj' in n or 'adamw_multi' in n:
                rows.append((n[:100], r['Counter_Name'], r['Counter_Value'], r.get('Grid_Size', ''), r.get('Workgroup_Size', '')))
    with open(os.path.join(out, tag + '_gemm_rows.csv'), 'w') as fo:
        w = csv.writer(fo); w.writerow(['kernel', 'counter', 'value', 'grid', 'workgroup']); w.writerows(rows)
    print(tag, len(rows), 'rows')
PY
echo "copy gpurun_out/prof_$rnd/{*_kernel_stats.csv,pmc_*_gemm_rows.csv,gemm_traffic.json} into profiles/$rnd/ (only gpurun_out/ travels back from the GPU box)"
