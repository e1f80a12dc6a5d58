#!/bin/bash
# PMC passes over a fused item-tile kernel (bench.py --roofline-only --model M --batch B --precision P): matrix pipe, LDS, texture
# addresser / L1 activity and the wave wait breakdown -- one rocprofv3 --pmc run per counter group (never combined with other traces).
# TA / TCP / TD counters go one per pass: four of them together exceed the hardware's slots (rocprofv3 aborts and does not exit).
# Usage: bash tools/pmc_fused.sh <model A|B> <batch> <f32|bf16|fp8> <tag>   -> gpurun_out/pmc_fused/<tag>.json
set -e
MODEL=$1; BATCH=$2; PREC=$3; TAG=$4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_fused/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "TA_TA_BUSY_sum" "TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TD_TD_BUSY_sum" "TCP_TOTAL_ACCESSES_sum" "TCP_TCC_READ_REQ_sum" \
            "TCP_PENDING_STALL_CYCLES_sum" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD"; do
  tag=$(echo $pass | tr ' ' '_' | cut -c1-60)
  timeout -k 10 180 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$tag -- python3 $ROOT/bench.py --roofline-only --model $MODEL --batch $BATCH --precision $PREC --quick > $OUT/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - $OUT <<'PY'
import csv, glob, collections, json, sys
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "fused" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in agg.items():
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    m["launches"] = max(len(v) for v in cs.values())
    gui = m.get("GRBM_GUI_ACTIVE")
    if gui:
        cyc = gui / 8.0  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m: m["mfma_busy_fraction"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
        m["kernel_cycles"] = cyc
    out[k] = m
json.dump(out, open(root + ".json", "w"), indent=1)
for k, m in out.items():
    print(k, json.dumps({kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in m.items()}))
PY
