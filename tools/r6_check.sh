#!/bin/bash
# Round 6, GPU box:  bash tools/r6_check.sh <run-tag> [skip-expr]
#   the GPU tests with the product library, the experiment tests with the experiments build (once), the headline line, the
#   line beyond the former size ceiling (config 6) and the two-sample batch.  A step that is killed at its limit ends the run.
TAG=${1:?usage: r6_check.sh <run-tag>}; SKIP=${2:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"; cd $R
step() {  # step <seconds> <log> <command...>
  local lim=$1 log=$2; shift 2
  timeout -k 10 $lim "$@" > $O/$log 2>&1; local rc=$?
  echo "$log exit=$rc"; tail -3 $O/$log | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed at its limit: stopping"; exit 1; fi
  return 0
}
step 900 tests.log python -m pytest tests -m gpu -q -p no:cacheprovider ${SKIP:+-k "$SKIP"}
export GPYREG_AMD_LIB=$R/gpyreg_amd/lib/libgpcore_exp.so
if [ -f "$GPYREG_AMD_LIB" ]; then
  step 600 tests_experiments.log python -m pytest tests -m "gpu and experiments" -q -p no:cacheprovider
fi
unset GPYREG_AMD_LIB
step 400 bench_cfg3.json python bench.py --steps 20 --warmup 5
step 300 bench_cfg3_S2.json python bench.py --steps 20 --warmup 5 --samples 2 --no-cpu-baseline
step 600 bench_cfg6.json python bench.py --config 6 --steps 3 --warmup 1
step 200 small_n_probe.txt python tools/small_n_probe.py
head -6 $O/small_n_probe.txt
