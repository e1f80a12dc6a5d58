#!/bin/bash
# with the workers on their own hardware queues: quarter-chip tiles, and more workers
set -o pipefail
O=gpurun_out/r04_prio2; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
run() { local label=$1; shift
  timeout -k 10 200 env "$@" > $O/o.out 2> $O/o.err
  echo "$label rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6))")" | tee -a $O/summary.txt
}
for prec in bf16 fp8; do
  args="--model C --batch 4096 --precision $prec"
  run "$prec 2x2 product rule" FR_X=0 python3 bench.py $args
  run "$prec 2x2 quarter-chip" FR_LP_GEMM_HALF=2 python3 bench.py $args
  for td in "3 1" "2 3" "3 2" "4 2"; do set -- $td
    run "$prec $1x$2" FR_X=0 python3 bench.py $args --threads $1 --depth $2
  done
  run "$prec 4x2 quarter-chip" FR_LP_GEMM_HALF=2 python3 bench.py $args --threads 4 --depth 2
  run "$prec 2x2 no prio" FR_STREAM_PRIO=0 python3 bench.py $args
done
