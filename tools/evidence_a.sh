#!/bin/bash
# Evidence of a round, part A (GPU box, repo root):  bash tools/evidence_a.sh <run-tag> <profiles-tag>
#   GPU tests -> smoke -> rocprofv3 kernel stats + the three PMC passes of the cfg3 bench -> profiles/<ptag>_* assembled ON
#   THE BOX (so that the traffic file exists before the tracked line is taken) -> the tracked bench line (it then quotes
#   the PMC traffic of its own code state).  Everything lands under gpurun_out/<run-tag>/ (profiles/ in its subdirectory).
set -o pipefail
TAG=${1:?usage: evidence_a.sh <run-tag> <profiles-tag>}; PTAG=${2:?profiles tag}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O/profiles"; cd $R
python3 tools/source_hash.py > $O/source.sha256
(timeout -k 10 700 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1; rc=$?; echo "pytest exit=$rc" >> $O/tests.log; tail -3 $O/tests.log; [ $rc -eq 0 ]) || exit 1
(timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; rc=$?; echo "smoke exit=$rc" >> $O/smoke.log; tail -2 $O/smoke.log; [ $rc -eq 0 ]) || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1; echo "rocprof stats exit=$?"
GPC_GROUPS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_g1 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_g1.log 2>&1; echo "rocprof(groups=1) exit=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc1 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc1.log 2>&1; echo "pmc1 exit=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $O/pmc2 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc2.log 2>&1; echo "pmc2 exit=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc3 -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc3.log 2>&1; echo "pmc3 exit=$?"
cd $R
python3 tools/assemble_profiles.py $TAG $PTAG > $O/assemble.log 2>&1; echo "assemble exit=$?"; tail -4 $O/assemble.log
# the tracked line: quotes profiles/<ptag>_traffic.json, written just now on this code state
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench exit=$?"
cp $O/bench.json profiles/${PTAG}_bench_cfg3.json
cp profiles/${PTAG}_* $O/profiles/
python3 -c "import json; d=json.load(open('$O/bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['frac_factor_section'], r['frac_wall'], r['traffic_code_state'], d['cpu_baseline']['value'], d['cpu_baseline']['grad_rel_err'])"
