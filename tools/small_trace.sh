#!/bin/bash
# kernel timeline of single evaluations (and of the 1024-point design) at a small N: small_trace.sh <tag> [N] [S ...]
TAG=${1:?tag}; N=${2:-100}; shift; shift; SS=${@:-1 1024}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for S in $SS; do
SMALL_N=$N SMALL_S=$S timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/tr$S -- python3 $R/tools/small_trace.py > $O/log$S.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/tr$S/*/*_kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
prev=None
out=open("$O/timeline_N${N}_S$S.txt","w")
for r in rows[-70:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    gap=(s-prev)/1e3 if prev else 0
    line="%-70s grid=%-8s dur=%7.2f us  gap=%7.2f us"%(r["Kernel_Name"].replace("void gpc::","")[:70], r["Grid_Size_X"]+"x"+r["Grid_Size_Y"], (e-s)/1e3, gap)
    print(line); out.write(line+"\n")
    prev=e
PY
done
