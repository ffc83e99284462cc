#!/bin/bash
# usage: pmc_gemm.sh OUTDIR "LAY PIPE HINT M N K" ...   -- three counter passes per configuration (own runs, kernel-trace only)
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace -d $out/${tag}_sq --output-format csv -- python3 $R/scratch/gemm_one.py $cfg > $out/${tag}_sq.log 2>&1 || exit 1
  timeout -k 10 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-trace -d $out/${tag}_tcc --output-format csv -- python3 $R/scratch/gemm_one.py $cfg > $out/${tag}_tcc.log 2>&1 || exit 1
  timeout -k 10 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace -d $out/${tag}_tcp --output-format csv -- python3 $R/scratch/gemm_one.py $cfg > $out/${tag}_tcp.log 2>&1 || exit 1
done
