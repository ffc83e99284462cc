cd $GRAFT_REPO_ROOT; O=gpurun_out
for l in lab_64x64 lab_64x64swp; do for s in "NN 2048 768 768" "NN 2048 768 3072" "NN 1600 768 2304" "NT 2048 768 768"; do timeout -k 10 60 python scratch/lab_run.py scratch/$l.so $s 2>&1 | grep "us/launch\|wg total\|k-step (all)" || exit 1; done; done > $O/lab_swp.log 2>&1
for l in lab_128x64 lab_128x64swp; do for s in "NT 2048 3072 768" "NN 2048 3072 768"; do timeout -k 10 60 python scratch/lab_run.py scratch/$l.so $s 2>&1 | grep "us/launch\|wg total\|k-step (all)" || exit 1; done; done >> $O/lab_swp.log 2>&1
cat $O/lab_swp.log
