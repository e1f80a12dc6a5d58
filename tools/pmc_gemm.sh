#!/bin/bash
# PMC passes over the Model-C batch-4096 FC1 GEMM (bench.py --model C --batch 4096 --precision P --quick): MFMA busy, LDS activity and
# conflicts, wave wait breakdown.  Usage: bash tools/pmc_gemm.sh <bf16|fp8> <tag> [env assignments...]  -> gpurun_out/pmc_gemm/<tag>.json
set -e
PREC=$1; TAG=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_gemm/$TAG
mkdir -p $OUT
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_BF16" \
            "TD_TD_BUSY_sum" "TA_TA_BUSY_sum" "TA_TOTAL_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-60)
  timeout -k 10 240 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$tag -- python3 $ROOT/bench.py --roofline-only --model C --batch 4096 --precision $PREC --quick > $OUT/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - $OUT <<'PY'
import csv, glob, collections, json, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "gemm" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in agg.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    m["launches"] = max(len(v) for v in cs.values())
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / 8.0
        m["kernel_cycles"] = cyc
        m["mfma_busy_fraction"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
        for c_, per in (("TD_TD_BUSY_sum", 256.0), ("TA_TA_BUSY_sum", 256.0), ("TCP_PENDING_STALL_CYCLES_sum", 256.0), ("TA_ADDR_STALLED_BY_TC_CYCLES_sum", 256.0)):
            if c_ in m: m[c_.replace("_sum", "") + "_fraction"] = m[c_] / (cyc * per)
        if "SQ_LDS_IDX_ACTIVE" in m: m["lds_active_fraction"] = m["SQ_LDS_IDX_ACTIVE"] / (cyc * 256.0)
    out[k] = m
json.dump(out, open(root + ".json", "w"), indent=1)
for k, m in out.items():
    print(k, json.dumps({kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in m.items()}))
PY
