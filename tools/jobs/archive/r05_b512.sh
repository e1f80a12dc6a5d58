#!/bin/bash
# round 5: what one rank of the 8-way table-sharded mode runs after the exchange -- Model-C's FC chain on 512 items: layer times and kernels
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for prec in f32 bf16 fp8; do
  echo "== $prec batch 512"
  timeout -k 10 300 python3 $R/bench.py --model C --batch 512 --precision $prec 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('   %.2f M inf/s  layers ms %s  conc %s kernels %s' % (j['value']/1e6, [round(x,4) for x in j.get('layer_launch_ms') or []], [round(x,2) for x in j.get('layer_concurrency') or []], j.get('layer_kernels')))
print('   ', {k: j[k] for k in ('kernel','dominant_kernel') if k in j}, j['roofline'].get('kernel_name'), j['roofline'].get('avg_launch_ms'))" || exit 1
done 2>&1 | tee $R/gpurun_out/r05_b512.txt
