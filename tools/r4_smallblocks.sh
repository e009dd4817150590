#!/bin/bash
# sweep of the 64-/128-tile threshold (GPC_SMALL_BLOCKS) at the per-GPU batch sizes of the configuration split
TAG=${1:?tag}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for S in 1 2 4 8 16; do
  for SB in 300 520 800 1100; do
    for MODE in fit nll; do
      GPC_SMALL_BLOCKS=$SB timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --samples $S --mode $MODE 2>/dev/null \
        | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('S=$S small_blocks=$SB $MODE ms_per_step=%.3f' % d['ms_per_step'])" | tee -a $O/smallblocks.txt
    done
  done
done
