#!/bin/bash
# Round 6, GPU box: measurements behind the pricing of split-k (S = 2 launch table) and the small-N latency floor.
TAG=${1:?tag}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for S in 1 2 4; do
  bash tools/step_breakdown.sh $TAG/s$S --samples $S > $O/breakdown_S$S.txt 2>&1
  python3 tools/launch_table.py $O/s$S/prof 40 > $O/launches_S$S.txt 2>&1
done
cd $R
timeout -k 10 200 python tools/small_n_probe.py > $O/small_default.txt 2>&1; echo "small default exit=$?"
HIP_FORCE_DEV_KERNARG=1 timeout -k 10 200 python tools/small_n_probe.py > $O/small_devkernarg1.txt 2>&1; echo "small devkernarg=1 exit=$?"
HIP_FORCE_DEV_KERNARG=0 timeout -k 10 200 python tools/small_n_probe.py > $O/small_devkernarg0.txt 2>&1; echo "small devkernarg=0 exit=$?"
head -6 $O/small_default.txt $O/small_devkernarg1.txt $O/small_devkernarg0.txt
cat $O/launches_S2.txt | head -60
