"""Per-DISPATCH rows of one GEMM class from a `rocprofv3 --pmc ... --kernel-trace` pass (tools/evidence_a.sh, pmc1):
grid, duration, MFMA-busy share -- which launches of a class lose the time.
usage: python tools/pmc_launches.py gpurun_out/<run>/pmc1 "<substring of the kernel name>" [more substrings ...]
MFMA utilisation of a dispatch = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x duration x clock): the counter sums the
busy cycles of all SIMDs; the clock is taken as SQ_BUSY_CYCLES / (duration x #SEs) when that counter is there, else 2.4 GHz
is NOT assumed -- the share is printed against SQ_BUSY_CYCLES per SIMD instead (the same normalisation as
assemble_profiles.py uses for the per-class figures)."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
pats = sys.argv[2:]
f = glob.glob(f"{d}/*/*counter_collection.csv")
assert len(f) == 1, f
rows = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    name = r["Kernel_Name"]
    if not any(p in name for p in pats):
        continue
    k = int(r["Dispatch_Id"])
    e = rows.setdefault(k, {"name": name.replace("void gpc::", "").split("(")[0], "grid": int(r["Grid_Size"]),
                            "wg": int(r["Workgroup_Size"]), "t0": int(r["Start_Timestamp"]), "t1": int(r["End_Timestamp"])})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
print(f"{'dispatch':>8} {'kernel':<58} {'blocks':>7} {'ms':>8} {'MFMA busy / (SQ busy x 4 SIMD)':>30} {'MOPS_F64':>12} {'flop/ms (TF/s)':>14}")
for k, e in rows.items():
    ms = (e["t1"] - e["t0"]) / 1e6
    busy = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    sqb = e.get("SQ_BUSY_CYCLES", 0.0)
    # SQ_BUSY_CYCLES: per-SE busy cycles summed over 32 SEs; MFMA busy: summed over 1024 SIMDs -> 32 SIMDs per SE
    util = busy / (sqb * 32.0) if sqb else float("nan")
    mops = e.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)
    # MOPS_F64 counts 512-flop units (one per 16x16x4 MFMA per wave / 4): flops = mops * 512
    tf = mops * 512.0 / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
    print(f"{k:8d} {e['name']:<58} {e['grid'] // e['wg']:7d} {ms:8.3f} {100 * util:29.1f}% {mops:12.3e} {tf:14.1f}")
