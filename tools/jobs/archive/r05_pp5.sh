#!/bin/bash
# round 5: fc_pp_gemm_kernel: more timing ablations / schedule variants (experiments build) -- FC1, two launches side by side, one process each
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export FR_LIB=$R/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for rep in 1 2; do
for prec in bf16 fp8; do
  for ab in 0 8 9 16 32 48; do
    echo "== $prec FR_PP_ABLATE=$ab"
    FR_PP_ABLATE=$ab timeout -k 10 300 python3 $R/bench.py --model C --batch 4096 --precision $prec --roofline-only 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('   layers ms %s  conc %s kernels %s' % ([round(x,4) for x in j['layer_launch_ms']], [round(x,2) for x in j.get('layer_concurrency')], j['layer_kernels'][:1]))" || exit 1
  done
done
done 2>&1 | tee $R/gpurun_out/r05_pp_ablate2.txt
