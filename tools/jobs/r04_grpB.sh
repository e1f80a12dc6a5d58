#!/bin/bash
# Model-B 1024 bf16: batches per launch of the persistent kernel (8 / 12 / 16 tiles per workgroup)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_grpB; mkdir -p $O
for rep in 1 2; do for g in 128 192 256; do for pb in "" "--per-bank"; do
  timeout -k 10 200 python3 bench.py --model B --batch 1024 --precision bf16 --group $g $pb > $O/o.out 2> $O/o.err
  echo "group=$g $pb rc=$? $(python3 -c "
import json
d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); rf=d['roofline']
print('%.2f M  kernel %.1f us frac %.3f' % (d['value']/1e6, 1e3*rf['avg_launch_ms'], rf['frac']))")" | tee -a $O/summary.txt
done; done; done
