#!/usr/bin/env python3
"""Per-kernel duration statistics of a rocprofv3 --kernel-trace CSV that `--stats` does not print: the MEDIAN (the stats file's AverageNs
includes the cold launches at the head of a run -- its MaxNs gives them away), p10 / p90, and, per stream, the SPAN per launch of the
kernel's longest uninterrupted run of back-to-back launches (first start -> last end, over the launches in it): what HIP events around
those launches measure.  avg > span means consecutive launches overlap (a kernel whose last workgroups trail lets its successor start).
usage: trace_kernel_median.py <kernel_trace.csv> [out.json]"""
import csv
import json
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
by = defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0") + "/" + r.get("Stream_Id", "0")))
out = {}
for name, ev in by.items():
    d = sorted(e[1] - e[0] for e in ev)
    n = len(d)
    rec = {"calls": n, "avg_us": sum(d) / n / 1e3, "median_us": d[n // 2] / 1e3, "p10_us": d[n // 10] / 1e3, "p90_us": d[min(n - 1, 9 * n // 10)] / 1e3, "max_us": d[-1] / 1e3}
    # back-to-back runs on one stream (hardware queue / stream id): consecutive launches of THIS kernel with less than 20 us between them --
    # queued launches follow each other within 0-1 us, a launch that waits for a small copy or for the host's pushes within ~10 us (HIP
    # events see those gaps too), a host synchronisation leaves 30-70 us before the next launch starts and ends the run (gap listings:
    # profiles/r05_experiments.md section 7).  bench.py times its launches in blocks and reports the MEDIAN block: so does this -- span per launch of
    # every run of >= 8 launches (first start -> last end over its launches), the median over the runs.
    runs = []
    streams = defaultdict(list)
    for e in sorted(ev):
        streams[e[2]].append(e)
    for L in streams.values():
        run = [L[0]]
        for a, b in zip(L, L[1:]):
            if b[0] - a[1] < 20000.0:
                run.append(b)
            else:
                runs.append(run)
                run = [b]
        runs.append(run)
    runs = [r_ for r_ in runs if len(r_) >= 8]
    # launches side by side on several streams (bench.py's stage-pipeline rows: one timed run of `reps` launches per worker, the streams' mean):
    # long runs (>= 30 launches) on more than one stream that overlap in time -> the launch-weighted MEAN of their spans is the like-for-like figure
    long_runs = [r_ for r_ in runs if len(r_) >= 30]
    side = len({r_[0][2] for r_ in long_runs}) > 1 and any(a_ is not b_ and a_[0][0] < b_[-1][1] and b_[0][0] < a_[-1][1] for a_ in long_runs for b_ in long_runs)
    if runs:
        sw = sorted(((max(e[1] for e in r_) - r_[0][0]) / len(r_) / 1e3, len(r_)) for r_ in runs)   # (span per launch, launches) per run
        spans = [x[0] for x in sw]
        rec["back_to_back_runs"] = len(runs)
        rec["back_to_back_launches"] = sum(len(r_) for r_ in runs)
        half, acc_ = rec["back_to_back_launches"] / 2.0, 0   # the median over the runs, every run weighted by its launches (a 10-launch warm-up
        for sp_, n_ in sw:                                   # run must not outvote a 60-launch timed one)
            acc_ += n_
            if acc_ >= half:
                rec["span_per_launch_us"] = sp_
                break
        rec["span_per_launch_min_us"], rec["span_per_launch_max_us"] = spans[0], spans[-1]
        if side:
            rec["side_by_side_streams"] = len({r_[0][2] for r_ in long_runs})
            rec["span_per_launch_us"] = sum((max(e[1] for e in r_) - r_[0][0]) / 1e3 for r_ in long_runs) / sum(len(r_) for r_ in long_runs)
        rec["avg_in_that_run_us"] = sum(e[1] - e[0] for r_ in runs for e in r_) / rec["back_to_back_launches"] / 1e3
    out[name] = rec
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
for name, rec in sorted(out.items(), key=lambda kv: -kv[1]["calls"] * kv[1]["avg_us"])[:12]:
    print("%-70s n=%5d avg %8.2f median %8.2f p10 %8.2f p90 %8.2f max %8.2f  span/launch %s" % (name[:70], rec["calls"], rec["avg_us"], rec["median_us"], rec["p10_us"], rec["p90_us"], rec["max_us"],
          "%8.2f (median of %d runs, %d launches, avg there %8.2f)" % (rec["span_per_launch_us"], rec["back_to_back_runs"], rec["back_to_back_launches"], rec["avg_in_that_run_us"]) if "span_per_launch_us" in rec else "-"))
