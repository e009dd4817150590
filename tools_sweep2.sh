#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-sweep2}; mkdir -p $O; cd $R
for ST in 0 1; do for G in 2 3 4 8; do
  r=$(GPC_STAGGER=$ST GPC_GROUPS=$G timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f fits/s  %.2f ms/step'%(d['value'],d['ms_per_step']))")
  echo "stagger=$ST groups=$G : $r" | tee -a $O/sweep.txt
done; done
