#!/bin/bash
# sweep of the dataflow graph's cut / priority / gate options (GPU box): bash tools/dag_sweep.sh <run-tag> "<N S grad>" opt-set ...
TAG=$1; shift; BASE=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for o in "$@"; do
  echo "== $BASE $o"
  GPC_DAG_STATS=1 timeout -k 10 300 python tools/dag_probe.py $BASE f64 5 $o 2>&1 | grep -a "stats\|wall ms\|identical: False\|aborts [1-9]" | awk '/stats/{s=$0; next} {print} END{if(s) print s}' | cut -c1-200
done
