#!/usr/bin/env python3
"""pmc_summarize.py <dir of passes> <key> <kernel-name substring> <json to update>

Averages the PMC passes of tools/pmc_passes.sh over the launches of ONE kernel and applies the gfx950 corrections of
MI355X_MICROARCH.md (HBM): FETCH_SIZE (KB) = TCC_EA0_RDREQ x 64 B counts 128-byte requests at 64 B -> doubled for the read side;
WRITE_SIZE (KB) is exact for 16-byte-per-lane stores.  The result is merged into the JSON under <key> (profiles/archive/r02_pmc.json is a
committed copy of it; bench.py reads `traffic_bytes_per_launch`, `l2_hit_rate`, `mfma_busy_fraction` from there)."""
import collections
import csv
import glob
import json
import os
import sys

root, key, kernel, out_json = sys.argv[1:5]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/*/*counter_collection.csv") + glob.glob(root + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if kernel in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
if not agg:
    sys.exit("no launches of a kernel matching %r under %s" % (kernel, root))
name = max(agg, key=lambda k: sum(len(v) for v in agg[k].values()))   # the instantiation with the most launches
cs = agg[name]
m = {c: sum(v) / len(v) for c, v in cs.items()}
e = {"kernel": name, "launches_averaged": {c: len(v) for c, v in cs.items()}, "raw": m}
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    e["fetch_bytes_corrected"] = 2.0 * m["FETCH_SIZE"] * 1024
    e["write_bytes"] = m["WRITE_SIZE"] * 1024
    e["traffic_bytes_per_launch"] = e["fetch_bytes_corrected"] + e["write_bytes"]
if "TCC_EA0_RDREQ_sum" in m:
    e["read_requests_per_launch"] = m["TCC_EA0_RDREQ_sum"]
    e["read_requests_32B_per_launch"] = m.get("TCC_EA0_RDREQ_32B_sum")
if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
    # busy cycles of all MFMA pipes / (active cycles of one XCD x 1024 SIMDs); counters are summed over the 8 XCDs
    e["mfma_busy_fraction"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
if "SQ_INSTS_VALU_MFMA_MOPS_F32" in m:
    e["mfma_f32_flops_per_launch"] = m["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512.0
if "TCC_HIT_sum" in m:
    e["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
allj = json.load(open(out_json)) if os.path.exists(out_json) else {}
allj[key] = e
json.dump(allj, open(out_json, "w"), indent=1)
print(key, json.dumps({k: v for k, v in e.items() if k != "raw"}))
