#!/bin/bash
# knob sweep on the GPU box: bash tools_sweep.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-sweep}; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1; echo "pytest exit=$?"; tail -2 $O/tests.log
for G in 1 2 4; do for SB in 768 1100 2300 4200 100000; do
  r=$(GPC_GROUPS=$G GPC_SMALL_BLOCKS=$SB timeout -k 10 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f fits/s  %.2f ms/step  dev %.2f ms'%(d['value'],d['ms_per_step'],d['roofline']['device_ms_per_step']))")
  echo "groups=$G small_blocks=$SB : $r" | tee -a $O/sweep.txt
done; done
