#!/bin/bash
# bash tools/evidence.sh <tag>: the secondary evidence of a round, all with cpu_baseline (BASELINE.md section 3 protocol):
# bench lines of cfg2 / cfg4 / cfg5 and cfg3 NLL-only, the latency table, a 2-rank rehearsal of the sharded bench (gloo,
# both ranks on this GPU), and the cfg4 rocprofv3 set (tools/prof_cfg4.sh).  Logs under gpurun_out/<tag>/.
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-ev}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O; cd $R
python3 tools/source_hash.py > $O/source.sha256
for c in 2 4 5; do
  timeout -k 10 500 python bench.py --config $c --steps 5 --warmup 2 > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err; echo "cfg$c exit=$?"
done
timeout -k 10 300 python bench.py --config 3 --steps 10 --warmup 2 --nll-only --no-cpu-baseline > $O/bench_cfg3_nll.json 2> $O/bench_cfg3_nll.err; echo "cfg3 nll exit=$?"
timeout -k 10 300 python tools/latency.py > $O/latency.txt 2>&1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo > $O/bench_cfg3_2ranks_gloo_one_gpu.json 2> $O/bench_2ranks.err; echo "2-rank rehearsal exit=$?"
for f in $O/bench_*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']; c=d.get('cpu_baseline') or {}
print('$f'.split('/')[-1], round(d['value'],2), d['unit'], round(d['ms_per_step'],3),'ms/step', round(r.get('achieved',0),1), r.get('unit'), round(r.get('frac',0),3), 'cpu', c.get('value'), 'err', c.get('nlz_rel_err'), c.get('grad_rel_err'))"; done
cat $O/latency.txt
bash tools/prof_cfg4.sh ${TAG}_cfg4
