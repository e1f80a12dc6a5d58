#!/bin/bash
# round 5: what would streaming fewer FC1 weight bytes per item buy the persistent bf16 kernel?  (the two-CU supertile, costed in r04_experiments.md section 7)
# TIMING ABLATION builds (wrong scores): every 3rd / every 2nd FC1 weight fragment is not loaded.  Model-B batch 1024 bf16, per-table and per-bank indices.
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
P=$R/gpu-fpga-recommendation-system_amd
for rep in 1 2; do
for lib in libfleetrec_exp.so libfleetrec_w1skip3.so libfleetrec_w1skip2.so; do
  for mode in "" "--per-bank"; do
    FR_LIB=$P/$lib python3 $R/bench.py --model B --batch 1024 --precision bf16 $mode 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('%-24s %-10s %.1f M inf/s   kernel %.1f us per launch (%s)' % ('$lib', '$mode' or 'per-table', j['value']/1e6, 1e3*j['roofline']['avg_launch_ms'], j['roofline']['kernel_name']))"
  done
done
done 2>&1 | tee $R/gpurun_out/r05_w1skip.txt
