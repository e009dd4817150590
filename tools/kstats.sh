#!/bin/bash
# bash tools/kstats.sh <tag> [pattern] [ENV=..]...: rocprofv3 kernel stats of a short bench run; prints rows matching the pattern
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; PAT=${2:-.}; shift 2; O=$R/gpurun_out/$TAG; mkdir -p $O
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $O/log.txt 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
python3 - $O/kernel_stats.csv "$PAT" <<'PY'
import csv,sys,re
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("%-90s calls %4s avg %10.1f us"%(r["Name"][:90], r["Calls"], float(r["AverageNs"])/1e3))
PY
rm -rf $O/prof
