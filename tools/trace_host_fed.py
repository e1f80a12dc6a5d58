#!/usr/bin/env python3
"""Reduce a rocprofv3 --kernel-trace --memory-copy-trace run of tools/host_fed_run.py to the question VERDICT r04 item 2 asks: where does the
host-fed request stream (H2D -> fused kernel -> D2H per block of 64 batches, cuda_server.c:460-461,494-495's hops) lose its few per cent
against the HBM-resident one?  Per phase (resident / host-fed, told apart by whether copies run): launches, kernel duration, kernels in
flight, time with NO kernel on the chip, and per stream the gap D2H(k) end -> kernel(k+1) start and H2D(k+1) end -> kernel(k+1) start.
Usage: trace_host_fed.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv> [kernel-name-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else "fr_fused_tile_m2_kernel"
kt = list(csv.DictReader(open(glob.glob(os.path.join(d, "*kernel_trace.csv"))[0])))
mt = list(csv.DictReader(open(glob.glob(os.path.join(d, "*memory_copy_trace.csv"))[0])))
K = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], r["Queue_Id"]) for r in kt if flt in r["Kernel_Name"])
C = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], "H2D" if "HOST_TO_DEVICE" in r["Direction"] else "D2H") for r in mt)
# phases: maximal runs of the kernel separated by > 20 ms of nothing
phases, cur = [], [K[0]]
for a, b in zip(K, K[1:]):
    if b[0] - a[1] > 20e6:
        phases.append(cur)
        cur = []
    cur.append(b)
phases.append(cur)


def pct(v, p):
    v = sorted(v)
    return v[min(len(v) - 1, int(p * len(v)))] if v else float("nan")


for ph in phases:
    if len(ph) < 200:
        continue
    n = len(ph)
    core = ph[n // 5: n - n // 10]      # steady part: drop the ramp and the drain
    t0, t1 = core[0][0], max(e[1] for e in core)
    copies = [c for c in C if t0 <= c[0] <= t1]
    kind = "host-fed" if len(copies) > len(core) // 2 else "resident"
    dur = [e[1] - e[0] for e in core]
    pts = sorted([(e[0], 1) for e in core] + [(e[1], -1) for e in core])
    hist, c_, last = defaultdict(int), 0, pts[0][0]
    for t, s in pts:
        hist[c_] += t - last
        c_ += s
        last = t
    tot = sum(hist.values())
    print("== %s phase: %d launches in the steady window of %.2f ms (%d streams, %d hardware queues)" % (kind, len(core), (t1 - t0) / 1e6, len({e[2] for e in core}), len({e[3] for e in core})))
    print("   launches per ms %.3f -> %.2f M inferences/s at 64 x 256 items per launch" % (len(core) / ((t1 - t0) / 1e6), len(core) * 16384 / ((t1 - t0) / 1e9) / 1e6))
    print("   kernel duration us: mean %.1f  p10 %.1f  p50 %.1f  p90 %.1f" % (sum(dur) / len(dur) / 1e3, pct(dur, .1) / 1e3, pct(dur, .5) / 1e3, pct(dur, .9) / 1e3))
    print("   kernels in flight: " + "  ".join("%d: %.1f %%" % (k, 100.0 * v / tot) for k, v in sorted(hist.items())) + "   (mean %.2f)" % (sum(k * v for k, v in hist.items()) / tot))
    by_q = defaultdict(list)
    for e in core:
        by_q[e[3]].append(e)
    qg = [b[0] - a[1] for L in by_q.values() for a, b in zip(L, L[1:])]
    print("   per hardware queue, end of a kernel -> start of the next one on that queue, us: p10 %.1f  p50 %.1f  p90 %.1f  mean %.1f" % (pct(qg, .1) / 1e3, pct(qg, .5) / 1e3, pct(qg, .9) / 1e3, sum(qg) / max(len(qg), 1) / 1e3))
    if kind == "host-fed":
        h2d = [c[1] - c[0] for c in copies if c[3] == "H2D"]
        d2h = [c[1] - c[0] for c in copies if c[3] == "D2H"]
        print("   copies: %d H2D of mean %.1f us (p90 %.1f), %d D2H of mean %.1f us (p90 %.1f)" % (len(h2d), sum(h2d) / max(len(h2d), 1) / 1e3, pct(h2d, .9) / 1e3, len(d2h), sum(d2h) / max(len(d2h), 1) / 1e3, pct(d2h, .9) / 1e3))
        # per stream: the order is H2D(k) -> kernel(k) -> D2H(k) -> H2D(k+1) ...
        ev = defaultdict(list)
        for e in core:
            ev[e[2]].append((e[0], e[1], "K"))
        for c in copies:
            ev[c[2]].append((c[0], c[1], c[3]))
        g_hk, g_kd, g_dk, g_dh = [], [], [], []
        for s, L in ev.items():
            L.sort()
            for a, b in zip(L, L[1:]):
                if a[2] == "H2D" and b[2] == "K":
                    g_hk.append(b[0] - a[1])
                if a[2] == "K" and b[2] == "D2H":
                    g_kd.append(b[0] - a[1])
                if a[2] == "D2H" and b[2] == "H2D":
                    g_dh.append(b[0] - a[1])
            ks = [x for x in L if x[2] == "K"]
            for a, b in zip(ks, ks[1:]):
                g_dk.append(b[0] - a[1])
        for name, g in (("H2D(k) end -> kernel(k) start", g_hk), ("kernel(k) end -> D2H(k) start", g_kd), ("D2H(k) end -> H2D(k+1) start", g_dh), ("kernel(k) end -> kernel(k+1) start, same stream", g_dk)):
            if g:
                print("   per stream, %-48s us: p10 %7.1f  p50 %7.1f  p90 %7.1f  mean %7.1f  (n=%d)" % (name, pct(g, .1) / 1e3, pct(g, .5) / 1e3, pct(g, .9) / 1e3, sum(g) / len(g) / 1e3, len(g)))
        # what runs beside a copy: is the chip idle while copies are the only thing moving?
        idle = hist.get(0, 0)
        print("   time with no kernel on the chip: %.2f %% of the window" % (100.0 * idle / tot))
    else:
        print("   time with no kernel on the chip: %.2f %% of the window" % (100.0 * hist.get(0, 0) / tot))
