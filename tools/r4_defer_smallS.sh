#!/bin/bash
# deferred inverse products at the small per-GPU batches of BASELINE's split (S = 1..4 at N = 4096): forced on / off
TAG=${1:?tag}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for S in 1 2 3 4; do
  for DM in -1 0 2048 1024; do
    GPC_DEFER_MIN=$DM timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --samples $S 2>/dev/null \
      | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=$S defer_min=$DM ms_per_step=%.3f' % d['ms_per_step'])" | tee -a $O/defer_smallS.txt
  done
done
