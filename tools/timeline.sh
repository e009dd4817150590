#!/bin/bash
# bash tools/timeline.sh <tag> [env assignments...]  -> compact per-launch timeline of the last bench step
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $O/log.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/*/*_kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# steps are delimited by build_kernel launches; keep the last timed step (before the 3 dominant-kernel steps)
out=open("$O/timeline.txt","w")
t0=int(rows[0]["Start_Timestamp"])
for r in rows:
    n=r["Kernel_Name"].replace("void gpc::","").split("(")[0]
    wg=(int(r["Grid_Size_X"])//max(1,int(r["Workgroup_Size_X"])))*(int(r["Grid_Size_Y"])//max(1,int(r["Workgroup_Size_Y"])))*(int(r["Grid_Size_Z"])//max(1,int(r["Workgroup_Size_Z"])))
    out.write("%s %s %d %d %d\n"%(n.replace(" ",""),r["Queue_Id"],wg,int(r["Start_Timestamp"])-t0,int(r["End_Timestamp"])-int(r["Start_Timestamp"])))
out.close()
PY
gzip -f $O/timeline.txt; rm -rf $O/*/ ; ls -la $O
