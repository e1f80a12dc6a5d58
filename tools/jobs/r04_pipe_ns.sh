#!/bin/bash
set -o pipefail
O=gpurun_out/r04_pipe_ns; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for shape in 14 15 16 23; do
  FR_GEMM_PIPE=$shape timeout -k 10 200 python3 bench.py --model C --batch 4096 --precision bf16 > $O/o.out 2> $O/o.err
  echo "bf16 FR_GEMM_PIPE=$shape rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6), [round(1e3*x,1) for x in d.get('layer_launch_ms')], d['roofline']['kernel_name'])")" | tee -a $O/summary.txt
done
for st in 2 3; do
  FR_GEMM_PIPE=0 FR_LP_GEMM_FC1_STAGES=$st timeout -k 10 200 python3 bench.py --model C --batch 4096 --precision bf16 > $O/o.out 2> $O/o.err
  echo "bf16 lp kernel rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6), [round(1e3*x,1) for x in d.get('layer_launch_ms')], d['roofline']['kernel_name'])")" | tee -a $O/summary.txt
done
