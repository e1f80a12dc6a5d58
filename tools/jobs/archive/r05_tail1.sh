#!/bin/bash
# round 5: the output layer folded into FC3's epilogue (fc_lp_gemm_out_kernel): parity, then A/B through the experiments build (FR_FC_TAIL=0: the two launches)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout -k 10 600 python -m pytest $R/tests/test_gpu_parity.py -x -q -m gpu -k "tiled_gemm or part_chip or coming_and_going or side_by_side or gemm_256 or table_sharded or bf16_chain or fp8_chain or config5_all or random_stage" 2>&1 | tail -5 || exit 1
export FR_LIB=$R/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for rep in 1 2; do
for prec in bf16 fp8; do
  for tail in 0 1; do
    echo "== $prec FR_FC_TAIL=$tail"
    FR_FC_TAIL=$tail python3 $R/bench.py --model C --batch 4096 --precision $prec 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('   %.2f M inf/s  layers ms %s  kernels %s' % (j['value']/1e6, [round(x,4) for x in j['layer_launch_ms']], j['layer_kernels']))"
  done
done
done 2>&1 | tee $R/gpurun_out/r05_tail_ab.txt
for prec in bf16 fp8; do
  echo "== per bank $prec"
  for tail in 0 1; do
  FR_FC_TAIL=$tail python3 $R/bench.py --model C --batch 4096 --precision $prec --per-bank 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('   FR_FC_TAIL=$tail %.2f M inf/s' % (j['value']/1e6))"
  done
done 2>&1 | tee -a $R/gpurun_out/r05_tail_ab.txt
