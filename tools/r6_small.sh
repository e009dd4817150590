#!/bin/bash
TAG=${1:?tag}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_small_path.py tests/test_gpu_boundary_cm.py tests/test_gpu_threads.py tests/test_gpu_fit.py tests/test_gpu_edge.py -q -p no:cacheprovider > $O/tests_small.log 2>&1; echo "tests exit=$?"; tail -3 $O/tests_small.log
timeout -k 10 200 python tools/small_n_probe.py > $O/small_poll.txt 2>&1; echo "probe exit=$?"
GPC_SMALL_POLL=0 timeout -k 10 200 python tools/small_n_probe.py > $O/small_nopoll.txt 2>&1; echo "probe(no poll) exit=$?"
head -12 $O/small_poll.txt; tail -1 $O/small_poll.txt; head -6 $O/small_nopoll.txt; tail -1 $O/small_nopoll.txt
timeout -k 10 300 python tools/fit_time.py > $O/fit_times.txt 2>&1; tail -8 $O/fit_times.txt
