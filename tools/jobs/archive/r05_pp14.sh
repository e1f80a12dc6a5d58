#!/bin/bash
# round 5: the 24-tile threshold of the 256 x 256 tile at chain width 4 on SMALLER batches (FC1 then has 32 / 64 tiles): FR_LP_GEMM_256_MIN=48 = the rule before
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export FR_LIB=$R/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for b in 1024 2048; do
for prec in bf16 fp8; do
  for mn in 24 48; do
    echo "== batch $b $prec FR_LP_GEMM_256_MIN=$mn"
    FR_LP_GEMM_256_MIN=$mn timeout -k 10 300 python3 $R/bench.py --model C --batch $b --precision $prec 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('   %.2f M inf/s  layers ms %s  conc %s kernels %s' % (j['value']/1e6, [round(x,4) for x in j['layer_launch_ms']], [round(x,2) for x in j.get('layer_concurrency')], j['layer_kernels']))" || exit 1
  done
done
done 2>&1 | tee $R/gpurun_out/r05_pp_small_batches.txt
