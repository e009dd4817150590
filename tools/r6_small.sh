#!/bin/bash
TAG=${1:?tag}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/tests.log 2>&1; echo "tests exit=$?"; tail -3 $O/tests.log
timeout -k 10 200 python tools/small_n_probe.py > $O/small_poll.txt 2>&1; echo "probe exit=$?"
head -10 $O/small_poll.txt; tail -1 $O/small_poll.txt
timeout -k 10 300 python tools/latency.py > $O/latency.txt 2>&1; cat $O/latency.txt
timeout -k 10 300 python tools/fit_time.py > $O/fit_times.txt 2>&1; tail -4 $O/fit_times.txt
