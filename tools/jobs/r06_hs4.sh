#!/bin/bash
# round 6: the persistent kernel on operand-type rows, timed the way the roofline legs are (native launch loop, HIP events), experiments build
mkdir -p gpurun_out
L=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for pb in "" "--per-bank"; do for v in 0 1 0 1; do
  FR_LIB=$L FR_FUSED_LP_ROWS=$v timeout -k 10 200 python bench.py --model B --batch 1024 --precision bf16 $pb 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=j.get('roofline') or {}
print('rows-image $v $pb: %.1f M inf/s; one stream %.1f us per launch (%s)' % (j['value']/1e6, 1e3*r.get('avg_launch_ms',0), r.get('kernel_name','')[:52]))" | tee -a gpurun_out/r06_lp_rows_hs_ab3.txt
done; done
