#!/bin/bash
# GEMM experiment flags on the GPU box (timing only for flags 2/4): bash tools_exp.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-exp}; mkdir -p $O; cd $R
for F in 0 1 2 4 6 3; do
  r=$(GPC_GROUPS=1 GPC_GEMM_FLAGS=$F timeout -k 10 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.1f fits/s  %.2f ms/step  factor %.2f ms'%(d['value'],d['ms_per_step'],d['roofline']['launch_ms']))")
  echo "flags=$F groups=1: $r" | tee -a $O/exp.txt
done
cd /tmp && export TMPDIR=/tmp
for F in 0 2 6; do
GPC_GROUPS=1 GPC_GEMM_FLAGS=$F timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f$F -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/prof_f$F.log 2>&1
done
