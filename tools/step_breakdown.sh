#!/bin/bash
# where one step goes: kernel time by class and the gaps, from a kernel trace of the LAST timed step.  usage: step_breakdown.sh <tag> <bench args...>
TAG=${1:?tag}; shift; R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline "$@" > $O/log.txt 2>&1
python3 - <<PY
import csv,glob,collections,re
f=glob.glob("$O/prof/*/*_kernel_trace.csv")[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
# steps begin with the upload kernel (xfer_kernel with the scaling) followed by a build kernel; find step starts = build_kernel launches
starts=[i for i,r in enumerate(rows) if "build_kernel" in r["Kernel_Name"]]
# the last TIMED step is the 4th evaluation (2 warm-up + 2): index 3
b=starts[3]; e=starts[4] if len(starts)>4 else len(rows)
seg=rows[b:e]
t0=int(seg[0]["Start_Timestamp"]); t1=max(int(r["End_Timestamp"]) for r in seg)
acc=collections.defaultdict(lambda:[0,0.0])
busy=[]
for r in seg:
    n=re.sub(r"\(.*","",r["Kernel_Name"].replace("void gpc::","").replace("void (anonymous namespace)::",""))
    g=int(r["Grid_Size_X"])//max(1,int(r["Workgroup_Size_X"]))*int(r["Grid_Size_Y"])
    key=n+("  [>=1024 blocks]" if g>=1024 else ("  [256..1023]" if g>=256 else "  [<256 blocks]"))
    acc[key][0]+=1; acc[key][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    busy.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"])))
busy.sort(); cov=0; cur_s,cur_e=busy[0]
for s_,e_ in busy[1:]:
    if s_>cur_e: cov+=cur_e-cur_s; cur_s,cur_e=s_,e_
    else: cur_e=max(cur_e,e_)
cov+=cur_e-cur_s
with open("$O/breakdown.txt","w") as out:
    def p(x):
        print(x); out.write(x+"\n")
    p("step span %.1f us, covered by at least one kernel %.1f us (gaps %.1f us), %d launches"%((t1-t0)/1e3,cov/1e3,(t1-t0-cov)/1e3,len(seg)))
    for k,(c,t) in sorted(acc.items(), key=lambda kv:-kv[1][1]):
        p("%-90s launches=%4d  sum=%9.1f us  avg=%7.1f us"%(k[:90],c,t,t/c))
PY
