#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for PAD in 0 24576; do for ST in 0 1; do for G in 1 2 4; do
  r=$(GPC_GEMM_PADLDS=$PAD GPC_STAGGER=$ST GPC_GROUPS=$G timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f fits/s  %.2f ms/step'%(d['value'],d['ms_per_step']))")
  echo "padlds=$PAD stagger=$ST groups=$G : $r"
done; done; done
