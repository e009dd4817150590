#!/bin/bash
# bash tools/kshape.sh <tag> [ENV=..]...: per (kernel, grid) average durations of one bench run under rocprofv3 --kernel-trace
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $O/log.txt 2>&1
python3 - $(find $O/prof -name "*kernel_trace.csv" | head -1) <<'PY'
import csv,sys,collections
g=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"].replace("void gpc::","").split("(")[0]
    if "gemm" not in n: continue
    wg=(int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))
    g[(n,wg)].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(g.items(), key=lambda kv:-sum(kv[1])):
    if sum(v)/7 > 150: print("%-50s grid %-12s n=%4d avg %9.1f us  total/step %8.1f us"%(k[0][:50],k[1],len(v),sum(v)/len(v),sum(v)/7))
PY
rm -rf $O/prof
