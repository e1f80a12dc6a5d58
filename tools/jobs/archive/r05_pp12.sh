#!/bin/bash
# round 5: fc_pp_gemm_n128_kernel (the phased-waves body on 128 x 256 tiles, 8-row sub-steps) for a lone worker (chain width 1): parity against the
# shipped kernels (fc_gemm_pipe_kernel bf16, fc_lp_gemm_kernel fp8), then A/B -- experiments build, FR_LP_GEMM_PP128
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export FR_LIB=$R/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so FR_CHECK_WIDTH=1
for v in 0 1; do
  FR_LP_GEMM_PP128=$v timeout -k 10 300 python3 $R/tools/experiments/gemm_pp_check.py /tmp/n128_$v.npz 2>&1 | tail -1 || exit 1
done
timeout 60 python3 $R/tools/experiments/gemm_pp_check.py /tmp/n128_0.npz /tmp/n128_1.npz 2>&1 | tee $R/gpurun_out/r05_pp_n128.txt
for rep in 1 2; do
for prec in bf16 fp8; do
  for v in 0 1; do
    echo "== $prec one worker FR_LP_GEMM_PP128=$v"
    FR_LP_GEMM_PP128=$v timeout -k 10 300 python3 $R/bench.py --model C --batch 4096 --precision $prec --threads 1 --depth 1 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('   %.2f M inf/s  layers ms %s  conc %s kernels %s' % (j['value']/1e6, [round(x,4) for x in j['layer_launch_ms']], [round(x,2) for x in j.get('layer_concurrency')], j['layer_kernels']))" || exit 1
  done
done
done 2>&1 | tee -a $R/gpurun_out/r05_pp_n128.txt
