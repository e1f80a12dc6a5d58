#!/bin/bash
# round 5: the GPU suite on the final GEMM rules, then the Model-C chain lines (product library)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout -k 10 1000 python -m pytest $R/tests -x -q -m gpu 2>&1 | tail -4 || exit 1
for rep in 1 2; do
for prec in bf16 fp8; do
    for bank in "" "--per-bank"; do
    echo "== $prec $bank"
    timeout -k 10 300 python3 $R/bench.py --model C --batch 4096 --precision $prec $bank 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('   %.2f M inf/s  layers ms %s  conc %s kernels %s' % (j['value']/1e6, [round(x,4) for x in j['layer_launch_ms']], [round(x,2) for x in j.get('layer_concurrency')], j['layer_kernels']))" || exit 1
    done
done
done 2>&1 | tee $R/gpurun_out/r05_pp_final_lines.txt
