#!/bin/bash
set -o pipefail
O=gpurun_out/r04_t256; mkdir -p $O
timeout -k 10 300 python3 tools/experiments/gemm_256_tile_check.py 2>&1 | tail -4 || exit 1
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for prec in bf16 fp8; do for t in 0 1; do
  FR_LP_GEMM_256=$t timeout -k 10 300 python3 bench.py --model C --batch 8192 --precision $prec > $O/o.out 2> $O/o.err
  echo "$prec batch 8192 tile256=$t rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6), [round(1e3*x,1) for x in d.get('layer_launch_ms')], d['roofline']['kernel_name'])")" | tee -a $O/summary.txt
done; done
