#!/bin/bash
# A/B of a GEMM kernel change: kernel tests, then bench (groups=1 and default)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fullsize.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -2
for rep in 1 2; do
for spec in "x 1" "x 2"; do set -- $spec
  GPC_GROUPS=$2 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); dk=d['roofline'].get('dominant_kernel',{})
print('groups=$2', round(d['value'],1),'fits/s', round(d['ms_per_step'],2),'ms/step; lauum', round(dk.get('achieved',0),1), 'TF', round(dk.get('launch_ms',0),3),'ms')"
done; done
