#!/bin/bash
# refresh the secondary numbers quoted in DESIGN.md section 6: bash tools/refresh_numbers.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:-nums}; mkdir -p $O; cd $R
for c in 2 4 5; do timeout -k 10 300 python bench.py --config $c --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg$c.json 2>/dev/null; done
timeout -k 10 300 python bench.py --config 3 --steps 10 --warmup 2 --nll-only --no-cpu-baseline > $O/bench_cfg3_nll.json 2>/dev/null
timeout -k 10 300 python tools/latency.py > $O/latency.txt 2>&1
timeout -k 10 300 python tools/predict_bench.py > $O/predict.txt 2>&1
timeout -k 10 300 python tools/design_bench.py > $O/design.txt 2>&1
for f in $O/bench_cfg*.json; do python3 -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); r=d['roofline']
print('$f'.split('/')[-1], round(d['value'],2), d['unit'], round(d['ms_per_step'],2),'ms/step', round(r.get('achieved',0),1), r.get('unit'), round(r.get('frac',0),3))"; done
tail -8 $O/latency.txt; cat $O/predict.txt $O/design.txt | grep -v Warn
