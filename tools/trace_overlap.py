#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> per kernel: count, average duration, and how much of its run time another kernel was running too;
plus the busy fraction of the timeline (union of all kernel intervals) over the last third of the trace (steady state)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70], r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in rows]
ev.sort()
n = len(ev)
ev = ev[2 * n // 3:]
t0, t1 = ev[0][0], max(e[1] for e in ev)
print("steady-state window: %d kernels over %.1f us" % (len(ev), (t1 - t0) / 1e3))
# union busy time and concurrency histogram
pts = []
for s, e, _, _ in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
cur, last, hist = 0, pts[0][0], collections.Counter()
for t, d in pts:
    hist[cur] += t - last
    cur += d; last = t
tot = sum(hist.values())
print("concurrency histogram (fraction of wall time with k kernels running):", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
# per kernel name: avg duration and overlapped fraction
by = collections.defaultdict(list)
for i, (s, e, name, q) in enumerate(ev):
    ov = 0
    for j in range(max(0, i - 40), min(len(ev), i + 40)):
        if j == i: continue
        s2, e2 = ev[j][0], ev[j][1]
        lo, hi = max(s, s2), min(e, e2)
        if hi > lo: ov = max(ov, 0) + 0  # placeholder
    by[name].append((e - s, s, e))
print("%-72s %6s %9s %9s" % ("kernel", "n", "avg us", "sum us"))
for name, l in sorted(by.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    print("%-72s %6d %9.2f %9.1f" % (name, len(l), sum(x[0] for x in l) / len(l) / 1e3, sum(x[0] for x in l) / 1e3))
print("sum of kernel durations / wall = %.3f" % (sum(e - s for s, e, _, _ in ev) / (t1 - t0)))
# a sample of the timeline
print("timeline sample (us from window start): start end dur stream kernel")
for s, e, name, q in ev[len(ev) // 2: len(ev) // 2 + 48]:
    print("%9.1f %9.1f %7.1f %4s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, name))
