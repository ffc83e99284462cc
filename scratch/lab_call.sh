cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout -k 10 300 python scratch/gemm_census.py cfg3_mcan_moe4 > $O/census_cfg3.log 2>&1 || { tail -5 $O/census_cfg3.log; exit 1; }
timeout -k 10 300 python scratch/gemm_census.py generative > $O/census_gen.log 2>&1 || { tail -5 $O/census_gen.log; exit 1; }
