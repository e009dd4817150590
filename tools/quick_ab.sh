#!/bin/bash
# bash tools/quick_ab.sh <tag> [full]: GPU tests (kernel tests only unless "full"), then cfg3 / cfg2 bench lines and the small-N latencies
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
if [ "$2" = full ]; then T="tests"; else T="tests/test_gpu_kernels.py tests/test_gpu_core_abi.py"; fi
timeout -k 10 900 python3 -m pytest $T -m gpu -x -q > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
BENCH_ARGS="--config 2" bash tools/ab_env.sh ${TAG}_c2 "A=1" "A=2"
bash tools/ab_env.sh $TAG "A=1" "A=2"
timeout -k 10 200 python3 tools/latency.py
