#!/bin/bash
# aux-stream gather x number of workers (threads x depth), Model-C 4096 bf16 / fp8, experiments build (FR_GATHER_AUX)
set -o pipefail
O=gpurun_out/r04_aux2; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for prec in bf16 fp8; do for td in "1 1" "1 2" "2 1" "2 2" "1 3"; do for aux in 0 1; do
  set -- $td
  FR_GATHER_AUX=$aux timeout -k 10 200 python3 bench.py --model C --batch 4096 --precision $prec --threads $1 --depth $2 > $O/o.out 2> $O/o.err
  echo "$prec threads=$1 depth=$2 aux=$aux rc=$? value=$(python3 -c "import json,sys; print('%.2f M' % (json.loads(open('$O/o.out').read().strip().splitlines()[-1])['value']/1e6))")" | tee -a $O/summary.txt
done; done; done
