#!/bin/bash
# Model-C 4096: 256 x 256 FC1 tiles on HALF the chip (128 workgroups), counting on two streams' FC1 launches running side by side
set -o pipefail
O=gpurun_out/r04_half256; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for prec in bf16 fp8; do for k in 1 2; do for q in "" 8; do
  [ -n "$q" ] && export GPU_MAX_HW_QUEUES=$q || unset GPU_MAX_HW_QUEUES
  FR_LP_GEMM_256=$k timeout -k 10 200 python3 bench.py --model C --batch 4096 --precision $prec > $O/o.out 2> $O/o.err
  echo "$prec LP_GEMM_256=$k hwq=${q:-default} rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6))")" | tee -a $O/summary.txt
done; done; done
