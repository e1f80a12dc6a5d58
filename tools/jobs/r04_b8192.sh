#!/bin/bash
# Model-C per item at batch 4096 vs 8192 (what larger GEMM tiles per compute unit could buy)
set -o pipefail
O=gpurun_out/r04_b8192; mkdir -p $O
for prec in bf16 fp8; do for b in 4096 8192; do
  timeout -k 10 300 python3 bench.py --model C --batch $b --precision $prec > $O/o.out 2> $O/o.err
  echo "$prec batch $b rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6), [round(1e3*x,1) for x in d.get('layer_launch_ms')], d.get('layer_kernels') or '')")" | tee -a $O/summary.txt
done; done
