"""summarise a tools/timeline.sh trace: per step, kernel-class busy time, union busy time, idle gaps."""
import gzip, sys, collections
rows = []
for line in gzip.open(sys.argv[1], "rt"):
    n, q, wg, t, d = line.split()
    rows.append((n, q, int(wg), int(t), int(d)))
# step starts: scale_x_kernel with the largest grid following a gap; use build_kernel launches of batch>=8
starts = [r[3] for r in rows if "scale_x_kernel" in r[0]]
# group scale_x launches that are within 1 ms into one step start
steps = []
for t in starts:
    if not steps or t - steps[-1] > 5e6:
        steps.append(t)
print("steps found:", len(steps))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo = steps[which]
hi = steps[which + 1] if which + 1 < len(steps) else 1 << 62
sel = [r for r in rows if lo <= r[3] < hi]
end = max(r[3] + r[4] for r in sel)
print(f"step {which}: {len(sel)} launches, span {(end-lo)/1e6:.3f} ms")
cls = collections.defaultdict(lambda: [0, 0])
for n, q, wg, t, d in sel:
    key = n
    if n.startswith("gemm_kernel"):
        key = n + (" big" if wg >= 512 else " small")
    cls[key][0] += 1
    cls[key][1] += d
for k, (c, d) in sorted(cls.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:60s} {c:5d} launches {d/1e6:8.3f} ms")
print(f"  sum of kernel durations {sum(r[4] for r in sel)/1e6:.3f} ms")
# union busy, and time when only 'small' launches (<256 WGs) are active
ev = []
for n, q, wg, t, d in sel:
    ev.append((t, 1, wg))
    ev.append((t + d, -1, wg))
ev.sort()
active = collections.Counter()
last = lo
busy = small_only = idle = 0
for t, s, wg in ev:
    dt = t - last
    tot = sum(active.values())
    if tot == 0:
        idle += dt
    else:
        busy += dt
        if sum(w * c for w, c in active.items()) < 256:
            small_only += dt
    active[wg] += s
    if active[wg] == 0:
        del active[wg]
    last = t
print(f"  union busy {busy/1e6:.3f} ms, idle gaps {idle/1e6:.3f} ms, of busy: <256 WGs in flight {small_only/1e6:.3f} ms")
if len(sys.argv) > 3:
    for n, q, wg, t, d in sel[: int(sys.argv[3])]:
        print(f"{(t-lo)/1e3:10.1f} us  +{d/1e3:8.1f}  q{q} wg={wg:6d} {n}")
