#!/bin/bash
TAG=${1:?tag}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/tests.log 2>&1; rc=$?; echo "tests exit=$rc"; tail -3 $O/tests.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python tools/latency.py > $O/latency.txt 2>&1; tail -7 $O/latency.txt
timeout -k 10 600 python tools/poll_soak.py 3000 > $O/poll_soak.txt 2>&1; echo "soak exit=$?"; tail -6 $O/poll_soak.txt
