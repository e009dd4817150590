"""How much of the latency-bound work of one sample group runs UNDER the other group's big launches?
usage: python tools/timeline_overlap.py gpurun_out/<tag>/timeline.txt.gz [step]"""
import gzip, sys, collections
rows = []
for line in gzip.open(sys.argv[1], "rt"):
    n, q, wg, t, d = line.split()
    rows.append((n, q, int(wg), int(t), int(d)))
starts = [r[3] for r in rows if "scale_x_kernel" in r[0]]
steps = []
for t in starts:
    if not steps or t - steps[-1] > 5e6:
        steps.append(t)
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo, hi = steps[which], (steps[which + 1] if which + 1 < len(steps) else 1 << 62)
sel = [r for r in rows if lo <= r[3] < hi]
end = max(r[3] + r[4] for r in sel)
print(f"step {which}: {len(sel)} launches, span {(end-lo)/1e6:.3f} ms, queues {sorted(set(r[1] for r in sel))}")
big = [(t, t + d, q) for n, q, wg, t, d in sel if wg >= 400 and "gemm" in n]
def under_big(t0, t1, q):
    tot = 0
    for a, b, qq in big:
        if qq == q:
            continue
        tot += max(0, min(t1, b) - max(t0, a))
    return tot
for pat in ("leaf3", "gemm_kernel"):
    cnt = dur = ov = 0
    per = collections.defaultdict(list)
    for n, q, wg, t, d in sel:
        if pat in n and wg < 400:
            cnt += 1; dur += d; ov += under_big(t, t + d, q)
            per[wg].append((d, under_big(t, t + d, q)))
    print(f"{pat}: {cnt} small launches, {dur/1e6:.3f} ms total, {ov/1e6:.3f} ms of it under the OTHER group's big GEMM launches")
    for wg, v in sorted(per.items()):
        alone = [d for d, o in v if o == 0]; und = [d for d, o in v if o > 0.5 * d]
        print(f"    grid {wg:5d}: {len(v):4d} launches; alone avg {sum(alone)/max(1,len(alone))/1e3:7.1f} us (n={len(alone)}), under a big launch avg {sum(und)/max(1,len(und))/1e3:7.1f} us (n={len(und)})")
# queue-level: busy time per queue and idle waiting
for q in sorted(set(r[1] for r in sel)):
    rs = sorted([r for r in sel if r[1] == q], key=lambda r: r[3])
    busy = sum(r[4] for r in rs)
    gaps = sum(max(0, rs[i + 1][3] - (rs[i][3] + rs[i][4])) for i in range(len(rs) - 1))
    print(f"queue {q}: {len(rs)} launches, busy {busy/1e6:.3f} ms, gaps between its launches {gaps/1e6:.3f} ms")
if len(sys.argv) > 3:
    for n, q, wg, t, d in sorted(sel, key=lambda r: r[3])[: int(sys.argv[3])]:
        print(f"{(t-lo)/1e3:10.1f} us  +{d/1e3:8.1f}  q{q} wg={wg:6d} {n}")
