#!/bin/bash
# Round 5: the dataflow graph (option "dag", csrc/dag.h) against the stream-ordered schedule -- bits, time per call, where the
# workgroups' time goes, and the critical path of one sample as it ran.   GPU box:  bash tools/r5_dag_evidence.sh <run-tag>
TAG=${1:?usage: r5_dag_evidence.sh <run-tag>}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
{
echo "# tools/dag_probe.py on one MI355X (round 5): GP.nll_batch through the stream-ordered schedule (plan.h) and through the dataflow"
echo "# graph (dag.h, option dag=1), same batch, wall clock of the call (min / median of 5) and hipEvent time of the device section;"
echo "# 'stats' = per-workgroup sums inside the graph's worker launch (100 MHz wall clock): life = pop (idle, polling) + acquire + run + complete."
echo "# args: N S grad dtype reps [option=value ...]"
for a in "2048 1 1 f64 5" "2304 4 0 f64 5" "2304 4 1 f64 5" "4096 1 1 f64 5" "4096 2 1 f64 5" "4096 4 1 f64 5" "4096 16 1 f64 5" "4096 16 0 f64 5" \
         "4096 16 1 f64 5 dag_gate=8" "4096 16 1 f64 5 dag_lauum=0" "4096 2 1 f64 5 dag_small_tiles=0" "4096 2 1 f64 5 dag_crit_pct=0" \
         "8192 1 1 f64 3" "8192 8 1 f64 3" "16384 1 1 f32 3"; do
  echo "== $a"
  GPC_DAG_LOG=1 GPC_DAG_STATS=1 timeout -k 10 300 python tools/dag_probe.py $a 2>&1 | grep -a "dag npad\|stats\|wall ms\|identical\|aborts" | awk '!seen[$0]++' | awk '/stats/{s=$0; next} {print} END{if(s) print s}'
done
} > $O/dag_probe.txt 2>&1
{
echo "# tools/dag_trace.py on one MI355X (round 5): per-task stamps of sample 0 inside the graph, joined with the graph; the critical path AS IT RAN"
for a in "4096 2" "2304 1"; do echo "== N S = $a"; DAG_TRACE_HOPS=30 timeout -k 10 200 python tools/dag_trace.py $a 2>&1 | grep -av "^\[gpcore\]"; done
} > $O/dag_trace.txt 2>&1
tail -5 $O/dag_probe.txt
