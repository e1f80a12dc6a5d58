#!/bin/bash
# round 5: fc_pp_gemm_kernel timing ablations (experiments build; wrong results by design except 0 and 8) -- FC1, two launches side by side
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export FR_LIB=$R/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
FR_LP_GEMM_PP=3 FR_PP_ABLATE=8 timeout -k 10 300 python3 $R/tools/experiments/gemm_pp_check.py /tmp/pp8.npz 2>&1 | tail -1
FR_LP_GEMM_PP=0 timeout -k 10 300 python3 $R/tools/experiments/gemm_pp_check.py /tmp/pp0.npz 2>&1 | tail -1
timeout 60 python3 $R/tools/experiments/gemm_pp_check.py /tmp/pp0.npz /tmp/pp8.npz 2>&1 | tee $R/gpurun_out/r05_pp_parity8.txt
for prec in bf16 fp8; do
  for ab in 0 8 1 2 3 4 5 6 7; do
    echo "== $prec FR_PP_ABLATE=$ab"
    FR_LP_GEMM_PP=3 FR_PP_ABLATE=$ab timeout -k 10 300 python3 $R/bench.py --model C --batch 4096 --precision $prec --roofline-only 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('   layers ms %s  conc %s kernels %s' % ([round(x,4) for x in j['layer_launch_ms']], [round(x,2) for x in j.get('layer_concurrency')], j['layer_kernels'][:1]))" || exit 1
  done
done 2>&1 | tee $R/gpurun_out/r05_pp_ablate.txt
