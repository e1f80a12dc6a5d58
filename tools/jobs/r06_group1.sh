#!/bin/bash
mkdir -p gpurun_out
for items in 131072 262144 131072 262144; do
  FR_BENCH_BF16_ITEMS=$items timeout -k 10 200 python bench.py --model B --batch 1024 --precision bf16 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=j.get('roofline') or {}
print('items/launch $items: %.1f M inf/s; one stream %.1f us per launch, frac %.4f' % (j['value']/1e6, 1e3*r.get('avg_launch_ms',0), r.get('frac',0)))" | tee -a gpurun_out/r06_bf16_group_ab.txt
done
