"""Every launch of the LAST timed step of a `rocprofv3 --kernel-trace` run of bench.py (tools/step_breakdown.sh writes
gpurun_out/<tag>/prof), in start order: start, duration, kernel, workgroups -- and, for launches of at least MIN_US, what
the step would gain if that launch ran at a given efficiency (the pricing of VERDICT r5 item 3).
usage: python tools/launch_table.py gpurun_out/<tag>/prof [min_us=50]"""
import csv
import glob
import re
import sys

d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
f = glob.glob(f"{d}/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "build_kernel" in r["Kernel_Name"] or "small_front_kernel" in r["Kernel_Name"]]
# bench.py --steps 2 --warmup 2: the 4th evaluation is the last timed one (the untimed 'alone' steps follow)
b = starts[3]
e = starts[4] if len(starts) > 4 else len(rows)
seg = rows[b:e]
t0 = int(seg[0]["Start_Timestamp"])
t1 = max(int(r["End_Timestamp"]) for r in seg)
print(f"last timed step: {len(seg)} launches, span {(t1 - t0) / 1e3:.1f} us")
small_n, small_t = 0, 0.0
for r in seg:
    n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void gpc::", "").replace("void (anonymous namespace)::", ""))
    wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    s, t = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if t >= min_us:
        print(f"  +{s:9.1f} us  {t:8.1f} us  {wg:6d} wg  {n[:80]}")
    else:
        small_n += 1
        small_t += t
print(f"  ({small_n} launches shorter than {min_us:.0f} us: {small_t:.1f} us in all)")
