#!/bin/bash
# bash tools/timeline_small.sh <tag> <N> [grad=1]: per-launch timeline of one single-sample evaluation at N (under rocprofv3)
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; N=$2; GRAD=${3:-1}; O=$R/gpurun_out/$TAG; mkdir -p $O
cat > $O/run.py <<PY
import os, sys
sys.path.insert(0, "$R")
import numpy as np
import bench
bench.CONFIGS[2] = dict(bench.CONFIGS[2], N=$N)
X, y, hyp = bench.synthetic_problem(2, 1)
gp = bench.make_gp(2, "f64")
gp.update(X_new=X, y_new=y, hyp=hyp, compute_posterior=False)
for _ in range(6):
    gp._GP__compute_nlZ(hyp[0], bool($GRAD), False)
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $O/run.py > $O/log.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/prof/*/*_kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# an evaluation = the transfer kernel before a build_kernel launch .. the transfer kernel before the next
idx=[i for i,r in enumerate(rows) if "build_kernel" in r["Kernel_Name"]]
lo=idx[-2]-1; hi=idx[-1]-1
t0=int(rows[lo]["Start_Timestamp"])
prev_end=t0
for r in rows[lo:hi]:
    n=r["Kernel_Name"].replace("void gpc::","").replace("gpc::","").split("(")[0]
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    wg=(int(r["Grid_Size_X"])//max(1,int(r["Workgroup_Size_X"])))*(int(r["Grid_Size_Y"])//max(1,int(r["Workgroup_Size_Y"])))
    print("%9.1f us  gap %6.1f  dur %7.1f  wg=%5d  %s"%((s-t0)/1e3,(s-prev_end)/1e3,(e-s)/1e3,wg,n[:70]))
    prev_end=e
print("evaluation span: %.1f us (first kernel start to the next evaluation's first kernel start: %.1f us)"%((prev_end-t0)/1e3,(int(rows[hi]["Start_Timestamp"])-t0)/1e3))
PY
rm -rf $O/prof
