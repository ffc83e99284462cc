cd $GRAFT_REPO_ROOT
bash scratch/r3_full.sh 12
