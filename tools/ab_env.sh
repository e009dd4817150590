#!/bin/bash
# bash tools/ab_env.sh <tag> "ENV1=a ENV2=b" "ENV1=c" ...   -> one bench line (fits/s, ms/step) per environment
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
i=0
for envs in "$@"; do
  i=$((i+1))
  env $envs timeout -k 10 120 python3 $R/bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline ${BENCH_ARGS} > $O/b$i.json 2> $O/b$i.err
  python3 - "$envs" $O/b$i.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[2]).read())
    dk=d["roofline"].get("dominant_kernel") or {}
    print("%-60s %8.2f fits/s  %7.3f ms/step  frac %.3f  lauum %.3f ms"%(sys.argv[1],d["value"],d["ms_per_step"],d["roofline"]["frac"],dk.get("launch_ms",0)))
except Exception as e:
    print(sys.argv[1],"FAILED",e)
PY
done
