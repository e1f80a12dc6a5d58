#!/bin/bash
# round 6: the persistent bf16 kernel's batch list on a copy stream of its own (FR_BLIST_COPY_STREAM=1, the product) vs on the worker's stream (0, round 5)
mkdir -p gpurun_out
L=$PWD/gpu-fpga-recommendation-system_amd
for v in 0 1 0 1; do
  for cfg in "A 256" "B 1024"; do
    set -- $cfg
    FR_LIB=$L/libfleetrec_exp.so FR_BLIST_COPY_STREAM=$v timeout -k 10 200 python bench.py --model $1 --batch $2 --precision bf16 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=j.get('roofline') or {}
print('copy_stream=$v  Model-$1 $2 bf16: %.1f M inf/s; one stream %.1f us per launch (%s)' % (j['value']/1e6, 1e3*r.get('avg_launch_ms',0), r.get('kernel_name','')[:52]))" | tee -a gpurun_out/r06_blist_copy_stream_ab.txt
  done
done
