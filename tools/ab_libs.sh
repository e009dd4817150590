#!/bin/bash
# A/B of prebuilt library variants: bash tools/ab_libs.sh <lib1.so> <lib2.so> ...   (each is copied over gpyreg_amd/lib/libgpcore.so)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
cp gpyreg_amd/lib/libgpcore.so /tmp/libgpcore_base.so
for lib in base "$@"; do
  [ $lib = base ] && cp /tmp/libgpcore_base.so gpyreg_amd/lib/libgpcore.so || cp $lib gpyreg_amd/lib/libgpcore.so
  timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1
  for spec in "0 1" "6 1" "0 2"; do set -- $spec
  GPC_GEMM_FLAGS=$1 GPC_GROUPS=$2 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); dk=d['roofline'].get('dominant_kernel',{})
print('$lib flags=$1 groups=$2', round(d['value'],1),'fits/s', round(d['ms_per_step'],2),'ms/step; lauum', round(dk.get('achieved',0),1), 'TF', round(dk.get('launch_ms',0),3),'ms')"
  done
done
cp /tmp/libgpcore_base.so gpyreg_amd/lib/libgpcore.so
