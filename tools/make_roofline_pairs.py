#!/usr/bin/env python3
"""Builds profiles/r04_roofline_pairs.json from one evidence job (tools/jobs/r04_evidence.sh): for every profiled leg, the HIP-event launch
time the profiled run printed on its own JSON line (stats_<leg>.log) beside the rocprofv3 average of the same kernel in that run's
kernel-stats CSV (<leg>_kernel_stats.csv) -- the same-run pairs tools/check_evidence.py checks to 3.5 %.
usage: python tools/make_roofline_pairs.py [gpurun_out/r04_evidence] [profiles/r04_roofline_pairs.json]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r04_evidence")
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "r04_roofline_pairs.json")


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
for leg in ("roofline", "gather_per_table_uniform", "gather_per_bank_uniform", "C4096_f32", "C4096_bf16", "C4096_fp8", "B1024_bf16", "B1024_bf16_per_bank", "B1024_f32",
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
                "rocprofv3_calls": int(hit[0]["Calls"]), "frac_same_run": rf["frac"]}
json.dump(out, open(dst, "w"), indent=1)
print("wrote %s (%d legs)" % (dst, len(out)))
