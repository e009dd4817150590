#!/bin/bash
# bash tools/prof_cfg4.sh <tag>: cfg4 (N=16384 D=20 RQ fp32, S=1) evidence: bench line, rocprofv3 kernel stats,
# and FETCH_SIZE / WRITE_SIZE / GRBM passes for the HBM-bound kernel-build regime (BASELINE.json configs[3])
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-cfg4}; O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O; python3 $R/tools/source_hash.py > $O/source.sha256
cd $R && timeout -k 10 300 python3 bench.py --config 4 --steps 5 --warmup 2 > $O/bench_cfg4.json 2> $O/bench_cfg4.err; echo "bench exit=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline > $O/prof.log 2>&1; echo "stats exit=$?"
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_cfg4.csv
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --kernel-trace --output-format csv -d $O/pmc_f -- python3 $R/bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_f.log 2>&1; echo "pmc fetch exit=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_w -- python3 $R/bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_w.log 2>&1; echo "pmc write exit=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_s -- python3 $R/bench.py --config 4 --steps 1 --warmup 1 --no-cpu-baseline > $O/pmc_s.log 2>&1; echo "pmc sq exit=$?"
python3 - <<PY
import csv,glob,collections,json
O="$O"
def load(t):
    f=glob.glob(f"{O}/{t}/*/*counter_collection.csv")
    return list(csv.DictReader(open(f[0]))) if f else []
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.defaultdict(collections.Counter)
for t in ("pmc_f","pmc_w","pmc_s"):
    for r in load(t):
        n=r["Kernel_Name"].replace("void gpc::","").split("(")[0]
        acc[n][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[n][r["Counter_Name"]]+=1
dur={}
for r in csv.DictReader(open(f"{O}/kernel_stats_cfg4.csv")):
    dur[r["Name"].replace("void gpc::","").split("(")[0]]=(float(r["AverageNs"]),int(r["Calls"]))
with open(f"{O}/pmc_summary_cfg4.txt","w") as out:
    out.write("cfg4 (N=16384, D=20, RQ, fp32, S=1): rocprofv3 --pmc passes of bench.py --config 4 --steps 1 --warmup 1 (5 evaluations per run), per-dispatch averages\n")
    for n,c in sorted(acc.items()):
        line=f"{n[:70]:70s} "+"  ".join(f"{k}={v/max(1,cnt[n][k]):.5g}" for k,v in sorted(c.items()))
        out.write(line+"\n")
    N=16384
    def one(prefixes):
        ks=[n for n in acc if any(n.startswith(p) for p in prefixes)]
        f=w=d=0.0
        for k in ks:
            c,m=acc[k],cnt[k]
            f+=2*1024*c.get("FETCH_SIZE",0)/max(1,m["FETCH_SIZE"]); w+=1024*c.get("WRITE_SIZE",0)/max(1,m["WRITE_SIZE"])
            d+=dur.get(k,(0,0))[0]*1e-9
        return ks,f,w,d
    for prefixes,alg,what in ((("build_kernel<float, 2, 0>","build_persist_kernel<float, 2, 0>"), N*(N+128)/2*4, "write of the lower half of A (fp32); head tiles + the persistent tail that runs under the first subtree"),
                              (("trace_kernel<float, 2, 0>",), N*(N+128)/2*4, "read of the lower half of K^-1 (fp32)")):
        ks,fetch,write,d=one(prefixes)
        if not ks or d<=0: continue
        s=(f"{' + '.join(ks)}: avg {d*1e3:.3f} ms per evaluation; algorithmic bytes {alg/1e9:.3f} GB ({what}) -> {alg/d/1e12:.2f} TB/s = {100*alg/d/8e12:.1f}% of 8 TB/s; "
           f"counter traffic 2 x FETCH_SIZE = {fetch/1e9:.3f} GB, WRITE_SIZE = {write/1e9:.3f} GB (ratio to algorithmic {(fetch+write)/alg:.2f})")
        print(s); out.write(s+"\n")
PY
# (the raw rocprofv3 directories stay under gpurun_out/ for re-analysis)
ls $O
