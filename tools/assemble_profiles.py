"""Copies the evidence of a tools/evidence_a.sh run (gpurun_out/<run>) into profiles/ under a round tag and
derives the PMC summary (MFMA utilisation, clock, memory-side traffic).  usage: assemble_profiles.py <run> <tag>"""
import collections, csv, glob, json, shutil, sys

run, tag = sys.argv[1], sys.argv[2]
O, P = f"gpurun_out/{run}", "profiles"


def one(pattern):
    """exactly one file may match: a second one means the directory holds more than one run"""
    f = glob.glob(pattern)
    if len(f) != 1:
        sys.exit(f"assemble_profiles: {len(f)} files match {pattern} (expected exactly 1): {f}\n"
                 "refusing to assemble evidence from a directory that holds more than one run")
    return f[0]


def names_of(path, col):
    return {r[col].replace("void gpc::", "").split("(")[0] for r in csv.DictReader(open(path))
            if "gpc::" in r[col] or r[col].startswith("gpc::")}


source = open(f"{O}/source.sha256").read().strip()
stats, stats_g1 = one(f"{O}/prof/*/*_kernel_stats.csv"), one(f"{O}/prof_g1/*/*_kernel_stats.csv")
pmc_files = {t: one(f"{O}/{t}/*/*counter_collection.csv") for t in ("pmc1", "pmc2", "pmc3")}
# every pass ran the same program on the same code: the kernel-name sets must agree (the per-run extras of the
# single-group stats run aside: it is the same bench with GPC_GROUPS=1)
sets = {"stats": names_of(stats, "Name"), "stats_g1": names_of(stats_g1, "Name"),
        **{t: names_of(f, "Kernel_Name") for t, f in pmc_files.items()}}
ref_set = sets["stats_g1"]
for k, v in sets.items():
    if v != ref_set:
        sys.exit(f"assemble_profiles: kernel-name sets differ between passes ({k} vs stats_g1): only in {k}: "
                 f"{sorted(v - ref_set)}; only in stats_g1: {sorted(ref_set - v)} -- the passes are not of one code state")
shutil.copy(stats, f"{P}/{tag}_kernel_stats_bench_cfg3.csv")
shutil.copy(stats_g1, f"{P}/{tag}_kernel_stats_bench_cfg3_groups1.csv")
# (the tracked bench line is taken AFTER this script has written the traffic file, so that it quotes the PMC traffic of
# its own code state: tools/evidence_a.sh copies it)
for line in open(f"{O}/prof_g1.log"):
    if line.startswith("{") and '"metric"' in line:
        open(f"{P}/{tag}_bench_cfg3_groups1_under_rocprof.json", "w").write(line)
        d = json.loads(line)
        print("groups=1 under rocprof: dominant launch_ms", d["roofline"].get("launch_ms"), "value", d["value"])
for r in csv.DictReader(open(f"{P}/{tag}_kernel_stats_bench_cfg3_groups1.csv")):
    if "true, true" in r["Name"]:
        print(r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e6, "ms avg", float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6)


def load(t):
    return list(csv.DictReader(open(pmc_files[t])))


out = open(f"{P}/{tag}_pmc_summary.txt", "w")
out.write(f"code state (tools/source_hash.py): {source}\n")
out.write("rocprofv3 --pmc passes on `bench.py --steps 1 --warmup 1` (cfg3: 1 warm-up + 1 timed step + 3 untimed steps with the mat-vecs after the W^T W launch);\n"
          "counters summed per kernel name over the run; the last block lists the W^T W launch (gemm_persist_kernel<double,true,true,128,4>) per dispatch\n")
per = {}
for t in ("pmc1", "pmc2", "pmc3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.OrderedDict()
    for r in load(t):
        n = r["Kernel_Name"].replace("void gpc::", "").split("(")[0]
        acc[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if "gemm_persist" in n and "true, true" in n:
            e = disp.setdefault(int(r["Dispatch_Id"]), {})
            e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
    for n, c in sorted(acc.items()):
        out.write(f"{t} {n[:60]:60s} " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(c.items())) + "\n")
    per[t] = disp
out.write("\nper dispatch, W^T W launch (16 samples each; the last three with the mat-vecs after it instead of under it):\n")
for t in per:
    for did, c in per[t].items():
        out.write(f"{t} dispatch {did}: " + "  ".join(f"{k}={v:.5g}" for k, v in sorted(c.items())) + "\n")
d1, d2, d3 = list(per["pmc1"].values())[-1], list(per["pmc2"].values())[-1], list(per["pmc3"].values())[-1]
grbm = d3["GRBM_GUI_ACTIVE"] / 8
busy = d1["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024
fetch, write = d2["FETCH_SIZE"] * 1024 * 2, d3["WRITE_SIZE"] * 1024
hit, miss = d2.get("TCC_HIT_sum", 0), d3.get("TCC_MISS_sum", 0)
txt = (f"\nderived, 16-sample W^T W launch: MFMA busy {busy:.4g} cycles per SIMD / {grbm:.4g} GPU cycles (GRBM_GUI_ACTIVE/8) = "
       f"{100*busy/grbm:.1f}% MFMA utilisation ({d1['SQ_VALU_MFMA_BUSY_CYCLES']/(4.016e11/2048):.1f} busy cycles per v_mfma_f64_16x16x4_f64, "
       f"1.961e8 MFMAs tile-exact);\nmemory-side traffic = 2 x FETCH_SIZE (gfx950 correction for 16 B/lane loads) + WRITE_SIZE = "
       f"{fetch/1e9:.2f} GB + {write/1e9:.2f} GB = {(fetch+write)/1e9:.2f} GB per launch (compulsory: 1.07 GB read of the lower half of W + "
       f"1.07 GB write of the lower half of the inverse, 16 samples); Infinity-Cache hits are included in FETCH_SIZE; "
       f"L2 hit rate TCC_HIT/(TCC_HIT+TCC_MISS) = {100*hit/max(1.0,hit+miss):.0f}%.\n")
out.write(txt)
print(txt)
# MFMA utilisation per GEMM class over the whole run: busy cycles / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), with the
# kernel time of the class per step from the stats pass.  (A CU-reserving launch runs on 192 of 256 CUs by design and a
# launch that shares the chip with another stream's kernels counts their time as its own: these are figures of the
# class AS SCHEDULED, not of the kernel alone.)
acc1 = collections.defaultdict(float)
acc3 = collections.defaultdict(float)
for r in load("pmc1"):
    if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
        acc1[r["Kernel_Name"].replace("void gpc::", "").split("(")[0]] += float(r["Counter_Value"])
for r in load("pmc3"):
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        acc3[r["Kernel_Name"].replace("void gpc::", "").split("(")[0]] += float(r["Counter_Value"])
ms = {}
for r in csv.DictReader(open(f"{P}/{tag}_kernel_stats_bench_cfg3.csv")):
    ms[r["Name"].replace("void gpc::", "").split("(")[0]] = float(r["TotalDurationNs"]) / 1e6 / 7.0  # 1 + 3 + 3 steps
out.write("\nMFMA utilisation per GEMM class (busy / (GRBM_GUI_ACTIVE/8 x 1024)), kernel time of the class per cfg3 step:\n")
for n in sorted(acc1):
    if "gemm" in n and acc3.get(n, 0) > 0 and acc1[n] > 0:
        line = f"  {n[:62]:62s} {100 * acc1[n] / (acc3[n] / 8 * 1024):5.1f} %   {ms.get(n, float('nan')):6.2f} ms per step"
        out.write(line + "\n")
        print(line)
# Per LAUNCH, the persistent 128-tile classes of one step (which launches of the 77.8 % class lose the
# time).  The PMC passes serialise kernels, so a CU-reserving launch (grid 512 + 96: U = T21 W11 of the top two levels)
# runs ALONE here on 192 of the 256 CUs -- its utilisation is quoted against the chip and against the CUs it may use.
rows_by_d = collections.OrderedDict()
for r in load("pmc1"):
    n = r["Kernel_Name"].replace("void gpc::", "").split("(")[0]
    if "gemm_persist_kernel" not in n:
        continue
    e = rows_by_d.setdefault(int(r["Dispatch_Id"]), {"name": n, "wg": int(r["Grid_Size"]) // int(r["Workgroup_Size"]),
                                                     "ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
out.write("\nper launch, persistent 128-tile GEMM classes (first step of the pmc1 pass; kernels run serialised under --pmc):\n"
          "  dispatch  kernel                                                  workgroups      ms   MFMA busy of the chip   of the CUs it may use   TFLOP/s (MOPS_F64 x 512 / ms)\n")
first = None
for did, e in rows_by_d.items():
    if "true, true" in e["name"]:
        first = did if first is None else first
    if first is not None and did > first:
        break  # one step: up to and including its W^T W launch
    sqb = e.get("SQ_BUSY_CYCLES", 0.0)
    util = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (sqb * 32.0) if sqb else float("nan")  # 32 SIMDs per shader engine
    reserved = e["wg"] > 2 * 256
    tf = e.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0) * 512.0 / (e["ms"] * 1e-3) / 1e12
    line = (f"  {did:8d}  {e['name'][:54]:54s}  {e['wg']:10d}  {e['ms']:6.3f}   {100 * util:20.1f} %   "
            f"{100 * util / (0.75 if reserved else 1.0):19.1f} %{' (192 CUs)' if reserved else '          '}   {tf:8.1f}")
    out.write(line + "\n")
    print(line)
nsteps = 5
tot_f = sum(float(r["Counter_Value"]) for r in load("pmc2") if r["Counter_Name"] == "FETCH_SIZE")
tot_w = sum(float(r["Counter_Value"]) for r in load("pmc3") if r["Counter_Name"] == "WRITE_SIZE")
json.dump({"dominant_kernel_traffic_bytes": fetch + write, "dominant_kernel_fetch_bytes_corrected": fetch,
           "dominant_kernel_write_bytes": write,
           "step_traffic_bytes": (tot_f * 2 + tot_w) * 1024 / nsteps,
           "step_note": "all kernels of one cfg3 NLL+grad step of 16 samples (run total of the PMC passes / 5 steps): 2 x FETCH_SIZE + WRITE_SIZE",
           "source_sha256": source,
           "source": f"profiles/{tag}_pmc_summary.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per MI355X_MICROARCH.md)"},
          open(f"{P}/{tag}_traffic.json", "w"), indent=1)
