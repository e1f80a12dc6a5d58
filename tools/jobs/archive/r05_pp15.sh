#!/bin/bash
# round 5: the minor-layer rule: batch 1024 / 2048 / 4096 chains on the product library + the GEMM-rule tests
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout -k 10 600 python -m pytest $R/tests/test_gpu_parity.py -x -q -m gpu -k "phased_waves or gemm_256 or chain_width or coming_and_going or side_by_side or tiled_gemm" 2>&1 | tail -3 || exit 1
for b in 1024 2048 4096; do
for prec in bf16 fp8; do
    echo "== batch $b $prec"
    timeout -k 10 300 python3 $R/bench.py --model C --batch $b --precision $prec 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('   %.2f M inf/s  layers ms %s  conc %s kernels %s' % (j['value']/1e6, [round(x,4) for x in j['layer_launch_ms']], [round(x,2) for x in j.get('layer_concurrency')], j['layer_kernels']))" || exit 1
done
done 2>&1 | tee $R/gpurun_out/r05_pp_minor_rule.txt
