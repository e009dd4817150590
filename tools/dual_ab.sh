#!/bin/bash
# bash tools/dual_ab.sh <tag>: tests, then cfg3 / cfg2 / small-N latency with GPC_DUAL=0 and 1
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout -k 10 400 python3 -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -x -q > $O/ktests.log 2>&1 || { tail -30 $O/ktests.log; exit 1; }
tail -3 $O/ktests.log
BENCH_ARGS="--config 2" bash tools/ab_env.sh ${TAG}_c2 "GPC_DUAL=0" "GPC_DUAL=1" "GPC_DUAL=0" "GPC_DUAL=1"
bash tools/ab_env.sh $TAG "GPC_DUAL=0" "GPC_DUAL=1" "GPC_DUAL=0" "GPC_DUAL=1"
for l in 0 1; do echo dual $l; GPC_DUAL=$l timeout -k 10 200 python3 tools/latency.py; done
