#!/bin/bash
# full GPU suite (keeps going after a failure: the log is what is wanted), then the default bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --maxfail=25 -s > $O/r3_tests_${1:-2}.log 2>&1; rc=$?
grep -E "passed|failed|FAILED|ERROR" $O/r3_tests_${1:-2}.log | tail -40
if [ $rc -ge 124 ]; then echo "pytest killed rc=$rc"; exit $rc; fi
exit 0
