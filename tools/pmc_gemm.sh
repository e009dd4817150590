#!/bin/bash
# PMC passes focused on the LDS/MFMA interplay of the lauum GEMM; bash tools/pmc_gemm.sh <tag> [flags...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GPC_GROUPS=1
for F in "$@"; do
export GPC_GEMM_FLAGS=$F
timeout -k 10 200 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/a$F -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/a$F.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_IFETCH SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/b$F -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $O/b$F.log 2>&1 || exit 1
done
python3 - <<PY
import csv,glob,collections
for d in sorted(glob.glob("$O/[ab]*/")):
    f=glob.glob(d+"*/*counter_collection.csv")
    if not f: print(d,"no counters"); continue
    acc=collections.defaultdict(float); n=collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if "gemm_kernel" in r["Kernel_Name"] and "true, true" in r["Kernel_Name"] and int(r["Grid_Size"])//int(r["Workgroup_Size"])==8448:
            acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
    print(d.split("/")[-2], " ".join("%s=%.4g(/%d)"%(k,v,n[k]) for k,v in sorted(acc.items())))
PY
rm -rf $O/[ab]*/
