#!/bin/bash
# round 4, first GPU pass: the new bench modes (self-launched ranks, predict) and the per-GPU batch sizes of the
# configuration split.  usage (GPU box, repo root): bash tools/r4_first.sh <tag>
set -o pipefail
TAG=${1:?usage: r4_first.sh <tag>}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd $R
python3 tools/source_hash.py > $O/source.sha256
timeout -k 10 500 python -m pytest tests/test_bench_launch.py -m gpu -q -x -p no:cacheprovider > $O/tests_launch.log 2>&1; echo "launch tests exit=$?"; tail -3 $O/tests_launch.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; echo "bench exit=$?"; cat $O/bench_cfg3.json
for S in 2 4 8; do
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --samples $S > $O/bench_cfg3_S$S.json 2>> $O/bench_cfg3.err; echo "S=$S exit=$?"
  python3 -c "import json;d=json.load(open('$O/bench_cfg3_S$S.json'));print('S=$S', d['value'], d['ms_per_step'], d['roofline']['frac_wall'])"
done
timeout -k 10 300 python bench.py --gpus 2 --backend gloo --steps 10 --warmup 3 > $O/bench_cfg3_2ranks_gloo_one_gpu.json 2> $O/bench_2r.err; echo "2 ranks exit=$?"; cat $O/bench_cfg3_2ranks_gloo_one_gpu.json
timeout -k 10 300 python bench.py --mode predict --steps 10 --warmup 3 > $O/bench_predict_cfg3.json 2> $O/bench_predict.err; echo "predict cfg3 exit=$?"; cat $O/bench_predict_cfg3.json
timeout -k 10 400 python bench.py --mode predict --config 5 --steps 5 --warmup 2 > $O/bench_predict_cfg5.json 2>> $O/bench_predict.err; echo "predict cfg5 exit=$?"; cat $O/bench_predict_cfg5.json
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_predict -- python3 $R/bench.py --mode predict --steps 5 --warmup 2 --no-cpu-baseline > $O/prof_predict.log 2>&1; echo "rocprof predict exit=$?")
find $O/prof_predict -name "*kernel_stats*" | head -2
