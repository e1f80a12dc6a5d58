#!/bin/bash
# Model-C 4096: half-chip tiles, part 2: FC2 on 128 x 128 tiles (LP_GEMM_HALF), per bank, and what fewer workers pay
set -o pipefail
O=gpurun_out/r04_half256b; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
run() { # label, env..., -- bench args
  local label=$1; shift
  timeout -k 10 200 env "$@" > $O/o.out 2> $O/o.err
  echo "$label rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6))")" | tee -a $O/summary.txt
}
for prec in bf16 fp8; do
  for pb in "" "--per-bank"; do
    for cfg in "1 0" "2 0" "2 1"; do set -- $cfg
      run "$prec $pb 256=$1 half=$2 2x2" FR_LP_GEMM_256=$1 FR_LP_GEMM_HALF=$2 python3 bench.py --model C --batch 4096 --precision $prec $pb
    done
  done
  for td in "1 1" "1 2" "4 1"; do set -- $td
    for k in 1 2; do
      run "$prec 256=$k threads=$1 depth=$2" FR_LP_GEMM_256=$k python3 bench.py --model C --batch 4096 --precision $prec --threads $1 --depth $2
    done
  done
done
