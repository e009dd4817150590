#!/bin/bash
# bash tools/timeline_big.sh <tag> <bench args...> [-- ENV=val ...]: per-launch timeline (big launches only) of the LAST timed bench step
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:?usage: timeline_big.sh <tag> <bench args...> [-- ENV=val ...]}; shift; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
ARGS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline "${ARGS[@]}" > $O/log.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/prof/*/*_kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "xfer_kernel" in r["Kernel_Name"]]
# a step = from the upload transfer kernel to the download one: pairs; take the second timed step
starts=idx[0::2]
lo=starts[2] if len(starts)>2 else starts[-1]
hi=starts[3] if len(starts)>3 else len(rows)
t0=int(rows[lo]["Start_Timestamp"])
small=0.0; nsmall=0
with open("$O/timeline.txt","w") as out:
    for r in rows[lo:hi]:
        n=r["Kernel_Name"].replace("void gpc::","").replace("gpc::","").split("(")[0]
        s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
        wg=(int(r["Grid_Size_X"])//max(1,int(r["Workgroup_Size_X"])))*(int(r["Grid_Size_Y"])//max(1,int(r["Workgroup_Size_Y"])))*(int(r["Grid_Size_Z"])//max(1,int(r["Workgroup_Size_Z"])))
        line="%9.1f us  dur %8.1f  q%-3s wg=%6d  %s"%((s-t0)/1e3,(e-s)/1e3,r["Queue_Id"],wg,n[:60])
        out.write(line+"\n")
        if (e-s)>60e3: print(line)
        else: small+=(e-s)/1e3; nsmall+=1
    end=max(int(r["End_Timestamp"]) for r in rows[lo:hi])
    print("step span %.1f us; %d launches under 60 us, %.1f us in all"%((end-t0)/1e3,nsmall,small))
PY
rm -rf $O/prof
