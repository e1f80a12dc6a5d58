#!/bin/bash
# round 5: a lone FC worker at 512 items (chain width 1): what one rank of the 8-way sharded step runs after the exchange
set -o pipefail
R=$GRAFT_REPO_ROOT
for prec in bf16 fp8; do
  echo "== $prec batch 512, one worker"
  timeout -k 10 300 python3 $R/bench.py --model C --batch 512 --precision $prec --threads 1 --depth 1 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('   %.2f M inf/s = %.1f us per batch; layers ms %s  conc %s kernels %s' % (j['value']/1e6, 512/j['value']*1e6, [round(x,4) for x in j.get('layer_launch_ms') or []], [round(x,2) for x in j.get('layer_concurrency') or []], j.get('layer_kernels')))" || exit 1
done 2>&1 | tee $R/gpurun_out/r05_b512b.txt
