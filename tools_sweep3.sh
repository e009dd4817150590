#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-sweep3}; mkdir -p $O; cd $R
GPC_DEFER=1 timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_core_abi.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
for DF in 0 1; do for G in 1 2; do for DM in 256 512 1024; do
  [ $DF = 0 -a $DM != 512 ] && continue
  r=$(GPC_DEFER=$DF GPC_DEFER_MIN=$DM GPC_GROUPS=$G timeout -k 10 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f fits/s  %.2f ms/step'%(d['value'],d['ms_per_step']))")
  echo "defer=$DF min=$DM groups=$G : $r" | tee -a $O/sweep.txt
done; done; done
for DF in 0 1; do r=$(GPC_DEFER=$DF timeout -k 10 120 python bench.py --config 2 --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f fits/s  %.3f ms/step'%(d['value'],d['ms_per_step']))"); echo "cfg2 defer=$DF: $r"; done
