cd $GRAFT_REPO_ROOT; O=gpurun_out
timeout -k 10 300 python scratch/norm_cov_tiny.py > $O/norm_cov_tiny.log 2>&1; grep -v amdgpu.ids $O/norm_cov_tiny.log | tail -8
bash scratch/r3_full.sh 11
