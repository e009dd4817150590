#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/gpu_run.sh <tag> [steps]   -- a FRESH tag per run: the box-side
# directory is wiped here, but gpurun MERGES the results into the local gpurun_out/<tag>, where an earlier run of the same
# tag would stay beside them (tools/assemble_profiles.py then refuses the directory)
# runs: GPU tests -> smoke -> bench -> rocprofv3 kernel trace of the bench; logs under gpurun_out/<tag>/
set -o pipefail
TAG=${1:-run}; STEPS=${2:-5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
# a run directory holds ONE run: a reused tag must not leave the previous run's CSVs beside the new ones
# (round 2's "r02f" set mixed two code states that way)
rm -rf $O; mkdir -p $O
cd $R
python3 tools/source_hash.py > $O/source.sha256
(timeout -k 10 700 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/tests.log 2>&1; rc=$?; echo "pytest exit=$rc" >> $O/tests.log; tail -3 $O/tests.log; [ $rc -eq 0 -o $rc -eq 1 ]) \
&& (timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; rc=$?; echo "smoke exit=$rc" >> $O/smoke.log; tail -2 $O/smoke.log; [ $rc -eq 0 -o $rc -eq 1 ]) \
&& (timeout -k 10 400 python bench.py --steps $STEPS --warmup 2 > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench exit=$rc"; cat $O/bench.json; tail -3 $O/bench.err; [ $rc -eq 0 -o $rc -eq 1 ]) \
&& (cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1; rc=$?; echo "rocprof exit=$rc"; tail -2 $O/prof.log; find $O/prof -name "*stats*" | head)
# same trace with one sample group: every lauum launch then carries all 16 samples and runs alone, so
# the kernel-stats average of gemm_persist_kernel<double,true,true,128,4> is the isolated duration that
# bench.py reports as roofline.dominant_kernel.launch_ms
(cd /tmp && export TMPDIR=/tmp && GPC_GROUPS=1 timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_g1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_g1.log 2>&1; echo "rocprof(groups=1) exit=$?")
# optional third arg "pmc": hardware-counter passes (own runs, kernel-trace only)
if [ "$3" = "pmc" ]; then
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 -L > $O/counters_list.txt 2>&1
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc1 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc1.log 2>&1; echo "pmc1 exit=$?"
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $O/pmc2 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc2.log 2>&1; echo "pmc2 exit=$?"
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc3 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc3.log 2>&1; echo "pmc3 exit=$?"
  find $O -name "*counter_collection*" | head
fi
# per-grid table of the dominant kernel from the trace (16-sample launches = the isolated measurement)
python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/prof_g1/*/*_kernel_trace.csv")
if f:
    g=collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "gemm_" in r["Kernel_Name"] and "true, true" in r["Kernel_Name"]:
            g[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"]), r["Grid_Size_Y"])].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
    with open("$O/lauum_by_grid.txt","w") as out:
        for k,v in sorted(g.items()):
            line="%s tiles=%d samples=%s launches=%d avg_ms=%.3f min_ms=%.3f"%(k[0],k[1],k[2],len(v),sum(v)/len(v)/1e6,min(v)/1e6)
            print(line); out.write(line+"\n")
PY
