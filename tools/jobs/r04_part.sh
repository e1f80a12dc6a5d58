#!/bin/bash
# part-chip tiles by worker count: parity, the divisor sweep at 2 x 2 workers, other batch sizes
set -o pipefail
O=gpurun_out/r04_part; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "half_chip or tiled_gemm_model_c or gemm_256 or committed_fc or sharded or two_contexts or driver" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest.log
[ $rc -ne 0 ] && exit 1
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
run() { local label=$1; shift
  timeout -k 10 200 env "$@" > $O/o.out 2> $O/o.err
  echo "$label rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6))")" | tee -a $O/summary.txt
}
for prec in bf16 fp8; do
  for part in 1 2 4 8 -1; do
    run "C 4096 $prec 2x2 part=$part" FR_LP_GEMM_PART=$part python3 bench.py --model C --batch 4096 --precision $prec
  done
  run "C 4096 $prec per-bank part=1" FR_LP_GEMM_PART=1 python3 bench.py --model C --batch 4096 --precision $prec --per-bank
  run "C 4096 $prec per-bank part=-1" FR_LP_GEMM_PART=-1 python3 bench.py --model C --batch 4096 --precision $prec --per-bank
  for b in 1024 2048 8192; do for part in 1 -1; do
    run "C $b $prec 2x2 part=$part" FR_LP_GEMM_PART=$part python3 bench.py --model C --batch $b --precision $prec
  done; done
done
