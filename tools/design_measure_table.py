"""DESIGN.md section 6's table from the committed bench lines of a round (profiles/<tag>_bench_*.json and the probe files):
python tools/design_measure_table.py r06   -> markdown on stdout."""
import json
import os
import re
import sys

tag = sys.argv[1]
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def L(name):
    with open(os.path.join(P, f"{tag}_{name}.json")) as fh:
        return json.load(fh)


def f3(x):
    return "--" if x is None else f"{x:.3f}"


def cpu(d):
    c = d.get("cpu_baseline")
    if not c:
        return "--"
    sec = c.get("seconds_per_eval") or c.get("seconds_per_call")
    s = f"{c['value']:.4g} {c['unit']} ({sec:.3g} s; {c['sample'].split(': ', 1)[-1].split(' (')[0]}; {c['cores']} BLAS threads)"
    if c.get("nlz_rel_err") is not None:
        s += f"; live parity nlZ {c['nlz_rel_err']:.1e}"
    if c.get("grad_rel_err") is not None:
        s += f", gradient {c['grad_rel_err']:.1e} per component"
    if c.get("mu_abs_err") is not None:
        s += f"; live parity mu {c['mu_abs_err']:.1e}, s2 {c['s2_rel_err']:.1e}"
    return s


def roof(d):
    r = d["roofline"]
    s = []
    if "launch_ms" in r and "lauum" in r.get("kernel", ""):
        s.append(f"W^T W in the pipeline {r['launch_ms']:.2f} ms = **{r['frac']:.3f}** of {r['peak']}")
        if r.get("alone"):
            s.append(f"alone {r['alone']['launch_ms']:.2f} ms = {r['alone']['frac']:.3f}")
    elif r.get("frac") is not None:
        s.append(f"{r['frac']:.3f} of {r['peak']} ({r['kernel'].split(':')[0][:40]})")
    if r.get("frac_factor_section") is not None:
        s.append(f"step {r['frac_factor_section']:.3f} by the factorization section")
    s.append(f"**{r['frac_wall']:.3f}** by the wall clock")
    return "; ".join(s)


rows = []
d = L("bench_cfg3")
rows.append(("**cfg3** N=4096 D=10 Matérn-5 S=16 fp64 NLL+grad (headline; `%s_bench_cfg3.json`)" % tag,
             f"**{d['value']:.1f} fits/s, {d['ms_per_step']:.2f} ms/step**", cpu(d), roof(d)))
d = L("bench_cfg3_nll")
rows.append(("cfg3 NLL only", f"{d['value']:.0f} evals/s, {d['ms_per_step']:.2f} ms per 16", cpu(d), roof(d) + " (N³/3 per evaluation)"))
d = L("bench_predict_cfg3")
rows.append(("**cfg3 `predict`**, 16 posteriors, M = 1000", f"{d['value']/1e6:.2f} M point-samples/s, {d['ms_per_step']:.2f} ms per call", cpu(d), roof(d)))
d = L("bench_predict_cfg5")
rows.append(("cfg5 `predict`, 64 posteriors of N = 8192, M = 1000", f"{d['value']/1e6:.2f} M point-samples/s, {d['ms_per_step']:.1f} ms", cpu(d), roof(d)))
d = L("bench_cfg2")
rows.append(("**cfg2** N=2048 D=5 SE S=1 fp64", f"{d['value']:.1f} fits/s, {d['ms_per_step']:.3f} ms", cpu(d), roof(d) + " (latency regime, §9 / §11)"))
d = L("bench_cfg4")
rows.append(("**cfg4** N=16384 D=20 RQ S=1 **fp32**", f"{d['value']:.1f} fits/s, {d['ms_per_step']:.2f} ms", cpu(d) + " (the reference has no fp32 path and no gradient at this size: §2)", roof(d)))
d = L("bench_cfg5")
rows.append(("**cfg5** N=8192 D=8 SE S=64 fp64", f"{d['value']:.1f} fits/s, {d['ms_per_step']:.1f} ms", cpu(d), roof(d)))
d = L("bench_cfg6")
rows.append(("**cfg6** (not a BASELINE configuration: beyond the former ceiling) N=32768 D=5 SE S=2 fp64", f"{d['value']:.2f} fits/s, {d['ms_per_step']:.0f} ms", cpu(d) + " (the reference's (N,N,6) gradient tensor alone is 52 GB)", roof(d)))
s = {k: L(f"bench_cfg3_S{k}") for k in (1, 2, 4, 8)}
s5 = L("bench_cfg5_S8")
rows.append(("the per-GPU batches of BASELINE's split on one GPU (`%s_bench_cfg3_S{1,2,4,8}.json`, `%s_bench_cfg5_S8.json`)" % (tag, tag),
             "cfg3 S = 1 / 2 / 4 / 8: **" + " / ".join(f"{s[k]['ms_per_step']:.2f}" for k in (1, 2, 4, 8)) + f" ms** per step; cfg5 S = 8: {s5['ms_per_step']:.1f} ms",
             "--", " / ".join(f3(s[k]['roofline']['frac_wall']) for k in (1, 2, 4, 8)) + f" and {f3(s5['roofline']['frac_wall'])} by the wall clock (§7, §9)"))
d = L("bench_cfg3_2ranks_gloo_one_gpu")
rows.append(("two self-launched ranks sharing ONE GPU (`bench.py --gpus 2 --backend gloo`)",
             f"the sharded path end to end, {d['config']['exchanges_per_step']:.1f} exchange per step, both splits and the rank-local yardstick in one record; the two processes time-slice the one device: a rehearsal of the record's shape, not a measurement", "--", "--"))
print("| workload | GPU | CPU oracle on the box's host | roofline |\n|---|---|---|---|")
for r in rows:
    print("| " + " | ".join(r) + " |")
