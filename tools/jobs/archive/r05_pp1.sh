#!/bin/bash
# round 5: fc_pp_gemm_kernel (the 256 x 256 GEMM tile with the SIMD's two waves in opposite phases): bit parity against fc_lp_gemm_kernel, then A/B
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export FR_LIB=$R/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for pp in 0 3 2; do
  FR_LP_GEMM_PP=$pp timeout -k 10 300 python3 $R/tools/experiments/gemm_pp_check.py /tmp/pp$pp.npz 2>&1 | tail -2 || exit 1
done
timeout 60 python3 $R/tools/experiments/gemm_pp_check.py /tmp/pp0.npz /tmp/pp3.npz 2>&1 | tee $R/gpurun_out/r05_pp_parity.txt || exit 1
timeout 60 python3 $R/tools/experiments/gemm_pp_check.py /tmp/pp0.npz /tmp/pp2.npz 2>&1 | tee -a $R/gpurun_out/r05_pp_parity.txt || exit 1
for rep in 1 2; do
for prec in bf16 fp8; do
  for pp in 0 3 2; do
    echo "== $prec FR_LP_GEMM_PP=$pp"
    FR_LP_GEMM_PP=$pp timeout -k 10 300 python3 $R/bench.py --model C --batch 4096 --precision $prec 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('   %.2f M inf/s  layers ms %s  conc %s kernels %s' % (j['value']/1e6, [round(x,4) for x in j['layer_launch_ms']], j.get('layer_concurrency'), j['layer_kernels'][:1]))" || exit 1
  done
done
done 2>&1 | tee $R/gpurun_out/r05_pp_ab.txt
