#!/bin/bash
# round 5: the product library after the body refactor: the 2 x 2-worker chain (unchanged kernels) and a lone worker (fc_pp_gemm_n128_kernel)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for rep in 1 2; do
for prec in bf16 fp8; do
  for w in "" "--threads 1 --depth 1"; do
    echo "== $prec $w"
    timeout -k 10 300 python3 $R/bench.py --model C --batch 4096 --precision $prec $w 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('   %.2f M inf/s  layers ms %s  conc %s kernels %s' % (j['value']/1e6, [round(x,4) for x in j['layer_launch_ms']], [round(x,2) for x in j.get('layer_concurrency')], j['layer_kernels']))" || exit 1
  done
done
done 2>&1 | tee $R/gpurun_out/r05_pp_final_lines2.txt
