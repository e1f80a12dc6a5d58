#!/bin/bash
set -o pipefail
O=gpurun_out/r04_fc2c; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for prec in bf16 fp8; do for rs in "8 2" "8 4" "16 2" "16 3"; do set -- $rs
  FR_LP_GEMM_ROWS=$1 FR_LP_GEMM_STAGES=$2 timeout -k 10 200 python3 bench.py --model C --batch 4096 --precision $prec > $O/o.out 2> $O/o.err
  echo "$prec rows=$1 stages=$2 rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6), [round(1e3*x,1) for x in d.get('layer_launch_ms')])")" | tee -a $O/summary.txt
done; done
