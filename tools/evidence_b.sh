#!/bin/bash
# Evidence of a round, part B: the other configurations and modes.  bash tools/evidence_b.sh <run-tag> <profiles-tag>
set -o pipefail
TAG=${1:?usage: evidence_b.sh <run-tag> <profiles-tag>}; PTAG=${2:?profiles tag}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O/profiles"; cd $R
python3 tools/source_hash.py > $O/source.sha256
b() { out=$1; shift; timeout -k 10 500 python bench.py "$@" > $O/profiles/${PTAG}_$out.json 2>> $O/bench.err; echo "$out exit=$?"; python3 -c "import json; d=json.load(open('$O/profiles/${PTAG}_$out.json')); print('   ', d['metric'], d['value'], d['ms_per_step'], d['roofline'].get('frac'), d['roofline'].get('frac_wall'))"; }
b bench_cfg3_nll --mode nll --steps 20 --warmup 5
b bench_cfg2 --config 2 --steps 50 --warmup 10
b bench_cfg5 --config 5 --steps 3 --warmup 1
b bench_cfg6 --config 6 --steps 3 --warmup 1
b bench_predict_cfg3 --mode predict --steps 20 --warmup 5
b bench_predict_cfg5 --mode predict --config 5 --steps 5 --warmup 2
for S in 1 2 4 8; do b bench_cfg3_S$S --samples $S --steps 20 --warmup 5 --no-cpu-baseline; done
b bench_cfg5_S8 --config 5 --samples 8 --steps 5 --warmup 2 --no-cpu-baseline
b bench_cfg3_2ranks_gloo_one_gpu --gpus 2 --backend gloo --steps 10 --warmup 3
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_predict -- python3 $R/bench.py --mode predict --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_predict.log 2>&1; echo "rocprof predict exit=$?")
cp $(find $O/prof_predict -name "*kernel_stats.csv" | head -1) $O/profiles/${PTAG}_kernel_stats_predict_cfg3.csv
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_nll -- python3 $R/bench.py --mode nll --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_nll.log 2>&1; echo "rocprof nll exit=$?")
cp $(find $O/prof_nll -name "*kernel_stats.csv" | head -1) $O/profiles/${PTAG}_kernel_stats_bench_cfg3_nll.csv
timeout -k 10 300 python tools/latency.py > $O/profiles/${PTAG}_latency.txt 2>/dev/null; echo "latency exit=$?"
timeout -k 10 300 python tools/small_n_probe.py > $O/profiles/${PTAG}_small_n_probe.txt 2>/dev/null; echo "small_n_probe exit=$?"
GPC_SMALL_PATH=0 timeout -k 10 300 python tools/small_n_probe.py 2>/dev/null | grep -v "N= 200\|N= 256\|N= 500" > $O/profiles/${PTAG}_small_n_probe_general_pipeline.txt
timeout -k 10 300 python tools/fit_time.py > $O/profiles/${PTAG}_fit_times.txt 2>/dev/null; echo "fit_time exit=$?"
bash tools/prof_cfg4.sh ${TAG}_cfg4 > $O/cfg4.log 2>&1; echo "cfg4 exit=$?"
cp $R/gpurun_out/${TAG}_cfg4/bench_cfg4.json $O/profiles/${PTAG}_bench_cfg4.json
cp $R/gpurun_out/${TAG}_cfg4/kernel_stats_cfg4.csv $O/profiles/${PTAG}_cfg4_kernel_stats.csv
cp $R/gpurun_out/${TAG}_cfg4/pmc_summary_cfg4.txt $O/profiles/${PTAG}_cfg4_pmc_summary.txt
tail -3 $O/profiles/${PTAG}_cfg4_pmc_summary.txt
