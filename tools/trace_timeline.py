#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel-trace CSV: per-kernel durations, per-stream gaps, and how many kernels
overlap in time (stream concurrency actually achieved).  Usage: trace_timeline.py <kernel_trace.csv> [name-filter]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
flt = sys.argv[2] if len(sys.argv) > 2 else None
ev = []
for r in rows:
    n = r["Kernel_Name"]
    short = n.split("(")[0].replace("void ", "")
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Stream_Id", r.get("Queue_Id", "0"))))
ev.sort()
# keep the steady-state part: from the first to the last kernel whose name matches the filter
if flt:
    idx = [i for i, e in enumerate(ev) if flt in e[2]]
    ev = ev[idx[len(idx) // 4]: idx[-1] + 1]  # skip the first quarter (warm-up / setup)
t0, t1 = ev[0][0], max(e[1] for e in ev)
dur = defaultdict(list)
for s, e, n, q in ev:
    dur[n].append(e - s)
print("window %.1f us, %d kernels" % ((t1 - t0) / 1e3, len(ev)))
for n, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    d.sort()
    print("  %-60s n=%5d  avg %7.2f us  p50 %7.2f  min %7.2f  max %7.2f  busy %5.1f%%" % (
        n[:60], len(d), sum(d) / len(d) / 1e3, d[len(d) // 2] / 1e3, d[0] / 1e3, d[-1] / 1e3, 100.0 * sum(d) / (t1 - t0)))
# concurrency histogram
pts = []
for s, e, n, q in ev:
    pts.append((s, 1))
    pts.append((e, -1))
pts.sort()
hist = defaultdict(int)
cur, last = 0, pts[0][0]
for t, d in pts:
    hist[cur] += t - last
    cur += d
    last = t
tot = sum(hist.values())
print("concurrency (kernels in flight): " + "  ".join("%d:%.1f%%" % (k, 100.0 * v / tot) for k, v in sorted(hist.items())))
print("average kernels in flight: %.2f" % (sum(k * v for k, v in hist.items()) / tot))
# per-stream gaps between consecutive kernels
by_q = defaultdict(list)
for s, e, n, q in ev:
    by_q[q].append((s, e, n))
gaps = []
for q, L in by_q.items():
    for a, b in zip(L, L[1:]):
        gaps.append(b[0] - a[1])
gaps.sort()
if gaps:
    print("gaps between consecutive kernels of one stream: p10 %.2f us  p50 %.2f us  p90 %.2f us (n=%d, %d streams)" % (
        gaps[len(gaps) // 10] / 1e3, gaps[len(gaps) // 2] / 1e3, gaps[9 * len(gaps) // 10] / 1e3, len(gaps), len(by_q)))
