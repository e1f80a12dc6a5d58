#!/bin/bash
# part-chip tiles: do the earlier negative experiments flip?  (fused tail, aux-stream gather, worker counts)
set -o pipefail
O=gpurun_out/r04_part2; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
run() { local label=$1; shift
  timeout -k 10 200 env "$@" > $O/o.out 2> $O/o.err
  echo "$label rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6))")" | tee -a $O/summary.txt
}
for prec in bf16 fp8; do
  a="--model C --batch 4096 --precision $prec"
  run "$prec base" FR_X=0 python3 bench.py $a
  run "$prec fc_tail" FR_FC_TAIL=1 python3 bench.py $a
  run "$prec gather_aux" FR_GATHER_AUX=1 python3 bench.py $a
  run "$prec gemm_gather" FR_GEMM_GATHER=1 python3 bench.py $a
  for td in "3 1" "4 1" "2 3" "3 2" "1 4"; do set -- $td
    run "$prec $1x$2" FR_X=0 python3 bench.py $a --threads $1 --depth $2
  done
  run "$prec base again" FR_X=0 python3 bench.py $a
done
