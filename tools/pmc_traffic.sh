#!/bin/bash
# HBM/fabric traffic and MFMA activity of the two roofline kernels from PMC counters, as MI355X_MICROARCH.md "HBM" prescribes:
# separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only, program directly after `--`.
# Run on the GPU box from the repo root:  bash tools/pmc_traffic.sh   -> gpurun_out/pmc_traffic/*.csv + summary JSON
set -e
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_traffic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum" \
            "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  tag=$(echo $pass | tr ' ' '_')
  rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$tag -- python3 $ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-zipf > $OUT/$tag.log 2>&1
done
cd $ROOT
python3 tools/pmc_summarize.py $OUT > $OUT/summary.json
cat $OUT/summary.json
