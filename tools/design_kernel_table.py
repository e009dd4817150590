"""The per-kernel table of DESIGN.md section 5, generated from the committed rocprofv3 summaries so that it cannot go stale
(VERDICT r4 item 7).  usage: python tools/design_kernel_table.py <tag> [--write]
reads  profiles/<tag>_kernel_stats_bench_cfg3_groups1.csv   (bench.py --steps 3 --warmup 1, one sample group: 1 warm-up + 3 timed + 3
                                                             untimed 'alone' steps = 7 steps of cfg3: N = 4096, D = 10, 16 samples)
       profiles/<tag>_pmc_summary.txt                       (MFMA utilisation per GEMM class, if present)
       profiles/<tag>_cfg4_kernel_stats.csv                 (cfg4, fp32, if present)
--write replaces the block between the markers '<!-- kernel-table:begin -->' and '<!-- kernel-table:end -->' in DESIGN.md."""
import csv, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
STEPS = 7
N, S, D = 4096, 16, 10
HBM = 8.0e12


def rows(path):
    out = []
    for r in csv.DictReader(open(path)):
        name = r["Name"].replace("void gpc::", "").replace("gpc::", "").split("(")[0]
        out.append((name, int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"])))
    return out


def util(tagname):
    """MFMA utilisation per kernel name from the pmc summary's 'per GEMM class' lines, if the file has them"""
    u = {}
    p = os.path.join(ROOT, "profiles", f"{tagname}_pmc_summary.txt")
    if os.path.exists(p):
        for line in open(p):
            m = re.match(r"\s+(gemm\S.*?>)\s+([\d.]+) %\s+([\d.]+) ms per step", line)
            if m:
                u[m.group(1).replace(" ", "")] = float(m.group(2))
    return u


ROLE = [
    ("gemm_persist_kernel<double, true, true", "W^T W = (K + sn2 I)^-1, all samples in one persistent launch", "fp64 MFMA", "F = S N^3 / 3"),
    ("gemm_persist_kernel<double, false, true", "U = T21 W11 (three CU-reserving launches on 192 of 256 CUs: 96.7 % / 85.6 % MFMA-busy on THOSE CUs, per-launch table in the pmc summary) and W21 = -W22 U of the root (whole chip: 96.8 %)", "fp64 MFMA", ""),
    ("gemm_persist_kernel<double, false, false", "T21 = A21 W11^T and the syrk updates of the top levels", "fp64 MFMA", ""),
    ("gemm_kernel<double, false, false, 64", "the same products at the 1024 / 512 levels as 64-tile launches", "fp64 MFMA (L2-bound in practice)", ""),
    ("gemm_kernel<double, false, true, 64", "inverse products of those levels", "fp64 MFMA (L2-bound in practice)", ""),
    ("gemm_dual_kernel", "syrk + U of a node in one grid (levels <= 512)", "latency", ""),
    ("leaf5_kernel", "128 x 128 Cholesky + inverse, 32 per sample", "latency: 128 dependent pivots", "2/3 128^3 flop per leaf"),
    ("trace_kernel", "gradient contraction sum_ij Q_ij dK_ij, dK recomputed from the inputs", "HBM by bytes (N^2/2 w read); fp64 VALU in practice", "bytes"),
    ("build_persist_kernel", "covariance build, rows >= 1024 (side stream)", "HBM by bytes (N^2/2 w written); fp64 VALU in practice", "bytes_build"),
    ("build_kernel", "covariance build, first 1024 rows", "HBM by bytes", ""),
    ("trmv_low_kernel", "z = W r under the W^T W launch (one wave of <= 32 VGPRs per SIMD beside the GEMM)", "HBM (issue-starved beside the GEMM)", "bytes"),
    ("trmv_t_part_low_kernel", "alpha = W^T z under the W^T W launch", "HBM (issue-starved beside the GEMM)", "bytes"),
    ("trmv_kernel", "z = W r alone (the 3 'alone' steps of bench.py)", "HBM", "bytes"),
    ("trmv_t_part_kernel", "alpha = W^T z alone", "HBM", "bytes"),
    ("xfer_kernel", "gathered upload / download of the call's small arrays", "PCIe latency", ""),
    ("grad_tail_kernel", "reduction of the tile partials, mean / noise gradient products", "latency", ""),
]


def table(path, title):
    rs = rows(path)
    total = sum(t for _, _, t, _ in rs) / STEPS / 1e6
    u = util(tag)
    lines = [f"*{title}: `{os.path.relpath(path, ROOT)}`, {STEPS} steps; kernel time {total:.2f} ms per step (sum of durations: launches overlap on two streams).*", "",
             "| kernel (rocprofv3 name) | role | launches / step | average | ms / step | bound | figure |", "|---|---|---|---|---|---|---|"]
    seen = set()
    for key, role, bound, fig in ROLE:
        for name, calls, tot, avg in rs:
            if name.startswith(key) and name not in seen:
                seen.add(name)
                per = tot / STEPS / 1e6
                extra = ""
                w = 8
                if fig == "bytes":
                    b = S * N * N / 2 * w
                    a = avg * 1e-9  # per LAUNCH (some of these run in only some of the 7 steps)
                    extra = f"{b / 1e9:.2f} GB / {a * 1e3:.3f} ms = {b / a / 1e12:.2f} TB/s = {100 * b / a / HBM:.0f} % of HBM peak"
                elif fig == "bytes_build":
                    b = S * (N * N - 1024 * 1024) / 2 * w
                    a = avg * 1e-9
                    extra = f"{b / 1e9:.2f} GB / {a * 1e3:.3f} ms = {b / a / 1e12:.2f} TB/s = {100 * b / a / HBM:.0f} % of HBM peak"
                elif fig.startswith("F ="):
                    f = S * N ** 3 / 3
                    extra = f"{fig} = {f / 1e9:.0f} GF / {avg / 1e6:.3f} ms = {f / (avg * 1e-9) / 1e12:.1f} TFLOP/s = {f / (avg * 1e-9) / 78.6e12:.3f} of 78.6"
                else:
                    extra = fig
                k2 = name.replace(" ", "")
                if k2 in u:
                    extra = (extra + "; " if extra else "") + f"MFMA utilisation {u[k2]:.1f} %"
                lines.append(f"| `{name}` | {role} | {calls / STEPS:.1f} | {avg / 1e3:.1f} us | {per:.3f} | {bound} | {extra} |")
    rest = [(n, c, t, a) for n, c, t, a in rs if n not in seen and t / STEPS / 1e6 >= 0.005 and n and not n.startswith("mfma_peak")]
    for name, calls, tot, avg in rest:
        lines.append(f"| `{name}` | | {calls / STEPS:.1f} | {avg / 1e3:.1f} us | {tot / STEPS / 1e6:.3f} | | |")
    return "\n".join(lines)


out = table(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats_bench_cfg3_groups1.csv"),
            f"cfg3 (N = 4096, D = 10, Matern-5, 16 samples, fp64, NLL + gradient), round tag {tag}")
block = "<!-- kernel-table:begin -->\n" + out + "\n<!-- kernel-table:end -->"
if "--write" in sys.argv:
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    assert "<!-- kernel-table:begin -->" in s and "<!-- kernel-table:end -->" in s, "markers missing in DESIGN.md"
    s = re.sub(r"<!-- kernel-table:begin -->.*?<!-- kernel-table:end -->", lambda m: block, s, flags=re.S)
    open(p, "w").write(s)
    print("DESIGN.md section 5 table rewritten from", tag)
else:
    print(block)
