#!/usr/bin/env python3
"""Builds profiles/rNN_roofline_pairs.json from one evidence job (tools/jobs/r06_evidence.sh): for every profiled leg, the HIP-event launch
time the profiled run printed on its own JSON line (stats_<leg>.log) beside the rocprofv3 figures of the same kernel in that run: the
kernel-stats CSV's average (<leg>_kernel_stats.csv) and, from the same run's kernel trace (<leg>_kernel_durations.json, made by
tools/trace_kernel_median.py), the MEDIAN duration and the span per launch of the back-to-back run -- the profiler's own equivalent of the
HIP-event figure.  tools/check_evidence.py checks the same-run pairs.
usage: python tools/make_roofline_pairs.py [gpurun_out/r06_evidence] [profiles/r06_roofline_pairs.json]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r06_evidence")
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r06_roofline_pairs.json")


def line_of(leg):
    for ln in reversed(open(os.path.join(src, "stats_%s.log" % leg)).read().splitlines()):
        if ln.startswith("{") and '"metric"' in ln:
            return json.loads(ln)
    raise SystemExit("no JSON line in stats_%s.log" % leg)


def roof_of(leg, d):
    if leg == "roofline":
        return d["roofline"]
    if leg.startswith("gather_per_table_zipf"):
        return d["gather"].get("zipf_1.05") or d["gather"]   # (the compact line of a --gather-law zipf run carries the zipf leg as `gather`)
    if leg.startswith("gather_per_table"):
        return d["gather"]
    if leg.startswith("gather_per_bank"):
        return d["gather_per_bank"]
    return d["roofline"]   # single-configuration runs (--model / --precision) carry the leg as the line's own roofline


out = {}
for leg in ("roofline", "gather_per_table_uniform", "gather_per_bank_uniform", "C4096_f32", "C4096_bf16", "C4096_fp8", "C4096_bf16_per_bank", "C4096_fp8_per_bank", "B1024_bf16", "B1024_bf16_per_bank", "B1024_f32",
            "A256_bf16", "A256_fp8"):
    if not os.path.exists(os.path.join(src, "stats_%s.log" % leg)):
        continue
    rf = roof_of(leg, line_of(leg))
    key = rf["kernel_name"].split("(")[0].strip().rstrip(">")   # (a name noted with fewer template arguments than rocprofv3 prints still matches its prefix)
    hit = [r for r in csv.DictReader(open(os.path.join(src, "%s_kernel_stats.csv" % leg))) if key in r["Name"]]
    if not hit:
        raise SystemExit("%s: kernel %s not in the CSV" % (leg, key))
    live = 1e3 * rf["avg_launch_ms"]
    out[leg] = {"kernel": rf["kernel_name"], "hip_events_us_same_run": live, "rocprofv3_avg_us": float(hit[0]["AverageNs"]) / 1e3,
                "rocprofv3_calls": int(hit[0]["Calls"]), "rocprofv3_max_us": float(hit[0]["MaxNs"]) / 1e3, "frac_same_run": rf["frac"]}
    dj = os.path.join(src, "%s_kernel_durations.json" % leg)
    if os.path.exists(dj):
        for name, rec in json.load(open(dj)).items():
            if key in name:
                out[leg].update({"rocprofv3_median_us": rec["median_us"], "rocprofv3_p10_us": rec["p10_us"], "rocprofv3_p90_us": rec["p90_us"],
                                 "rocprofv3_span_per_launch_us": rec.get("span_per_launch_us"), "rocprofv3_avg_in_back_to_back_run_us": rec.get("avg_in_that_run_us"),
                                 "back_to_back_launches": rec.get("back_to_back_launches"), "back_to_back_runs": rec.get("back_to_back_runs"),
                                 "rocprofv3_span_per_launch_min_us": rec.get("span_per_launch_min_us"), "rocprofv3_span_per_launch_max_us": rec.get("span_per_launch_max_us")})
                break
json.dump(out, open(dst, "w"), indent=1)
print("wrote %s (%d legs)" % (dst, len(out)))
