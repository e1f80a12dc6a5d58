#!/bin/bash
# round 5: fc_pp_gemm_kernel: the fetching wave at raised priority (FR_PP_ABLATE=64, with / without the multiplying wave's: 80) -- experiments build
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export FR_LIB=$R/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for rep in 1 2; do
for prec in bf16 fp8; do
  for ab in 0 64 16 80; do
    echo "== $prec FR_PP_ABLATE=$ab"
    FR_PP_ABLATE=$ab timeout -k 10 300 python3 $R/bench.py --model C --batch 4096 --precision $prec 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('   %.2f M inf/s  layers ms %s  conc %s' % (j['value']/1e6, [round(x,4) for x in j['layer_launch_ms']], [round(x,2) for x in j.get('layer_concurrency')]))" || exit 1
  done
done
done 2>&1 | tee $R/gpurun_out/r05_pp_prio.txt
