#!/bin/bash
# round 5: FC2 (1024 -> 512 at batch 4096) on 32 tiles of 256 x 256 (fc_pp_gemm_kernel) instead of 64 of 128 x 256: FR_LP_GEMM_PART=8 (experiments build)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export FR_LIB=$R/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for rep in 1 2; do
for prec in bf16 fp8; do
  for part in -1 8; do
    for bank in "" "--per-bank"; do
    echo "== $prec $bank FR_LP_GEMM_PART=$part"
    FR_LP_GEMM_PART=$part timeout -k 10 300 python3 $R/bench.py --model C --batch 4096 --precision $prec $bank 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('   %.2f M inf/s  layers ms %s  conc %s kernels %s' % (j['value']/1e6, [round(x,4) for x in j['layer_launch_ms']], [round(x,2) for x in j.get('layer_concurrency')], j['layer_kernels']))" || exit 1
    done
  done
done
done 2>&1 | tee $R/gpurun_out/r05_pp_fc2_256.txt
