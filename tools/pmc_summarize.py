#!/usr/bin/env python3
"""Average the PMC passes of tools/pmc_traffic.sh per kernel and apply the gfx950 corrections of
MI355X_MICROARCH.md (HBM): FETCH_SIZE (KB) = TCC_EA0_RDREQ x 64 B counts 128-byte requests at 64 B -> doubled for the
read side; WRITE_SIZE (KB) is exact for 16-byte-per-lane stores."""
import collections
import csv
import glob
import json
import sys

root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in agg.items():
    if not any(x in k for x in ("gather_pack", "fr_pipeline_kernel<-1", "fr_fused_tile")):
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    e = {"launches_averaged": {c: len(v) for c, v in cs.items()}, "raw": m}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        e["fetch_bytes_corrected"] = 2.0 * m["FETCH_SIZE"] * 1024
        e["write_bytes"] = m["WRITE_SIZE"] * 1024
        e["traffic_bytes_per_launch"] = e["fetch_bytes_corrected"] + e["write_bytes"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        # rocprofv3's derived MfmaUtil: busy cycles of all MFMA pipes / (active cycles of one XCD x 1024 SIMDs); the counter
        # values are summed over the 8 XCDs, so GRBM_GUI_ACTIVE / 8 is the per-XCD figure
        e["mfma_busy_fraction"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
    if "SQ_INSTS_VALU_MFMA_MOPS_F32" in m:
        e["mfma_f32_flops_per_launch"] = m["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512.0
    if "TCC_HIT_sum" in m:
        e["l2_hit_rate"] = m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    out[k] = e
print(json.dumps(out, indent=1))
