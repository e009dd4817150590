#!/bin/bash
# bash tools/leaf_ab.sh <tag>: kernel tests, then cfg3 / cfg2 / small-N latency with the phase-ordered (3) and the pipelined (5) leaf
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_kernels.py -x -q > $O/ktests.log 2>&1 || { tail -30 $O/ktests.log; exit 1; }
tail -3 $O/ktests.log
bash tools/ab_env.sh $TAG "GPC_LEAF=3" "GPC_LEAF=5" "GPC_LEAF=3" "GPC_LEAF=5"
BENCH_ARGS="--config 2" bash tools/ab_env.sh ${TAG}_c2 "GPC_LEAF=3" "GPC_LEAF=5"
for l in 3 5; do echo leaf $l; GPC_LEAF=$l timeout -k 10 200 python3 tools/latency.py; done
