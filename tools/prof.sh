#!/bin/bash
# bash tools/prof.sh <tag> <bench args...>  -> kernel stats of one bench invocation
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py "$@" --no-cpu-baseline > $O/log.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if float(r['TotalDurationNs'])>2e4: print("%-60s calls %5s avg %9.1f us total %8.3f ms"%(r['Name'][:60].replace('void gpc::',''),r['Calls'],float(r['AverageNs'])/1e3,float(r['TotalDurationNs'])/1e6))
PY
