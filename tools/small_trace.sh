#!/bin/bash
TAG=${1:?tag}; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for S in 1 1024; do
SMALL_S=$S timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/tr$S -- python3 $R/tools/small_trace.py > $O/log$S.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/tr$S/*/*_kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
# last 2 NLL+grad calls and the NLL ones before: print the last 40 kernels
prev=None
out=open("$O/timeline_S$S.txt","w")
for r in rows[-60:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    gap=(s-prev)/1e3 if prev else 0
    line="%-70s grid=%-8s dur=%7.2f us  gap=%7.2f us"%(r["Kernel_Name"].replace("void gpc::","")[:70], r["Grid_Size_X"]+"x"+r["Grid_Size_Y"], (e-s)/1e3, gap)
    print(line); out.write(line+"\n")
    prev=e
PY
done
