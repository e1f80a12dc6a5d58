#!/bin/bash
# half-chip tiles as the product rule (second worker on the context): parity tests, then the bench rows with the PRODUCT library
set -o pipefail
O=gpurun_out/r04_half256c; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "half_chip or tiled_gemm_model_c or gemm_256 or committed_fc or sharded or two_contexts" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -8 $O/pytest.log
[ $rc -ne 0 ] && exit 1
for prec in bf16 fp8; do for pb in "" "--per-bank"; do
  timeout -k 10 200 python3 bench.py --model C --batch 4096 --precision $prec $pb > $O/o.out 2> $O/o.err
  echo "$prec $pb rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6))")" | tee -a $O/summary.txt
done; done
for b in 2048 8192; do for prec in bf16 fp8; do for h in -1 0; do
  FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so FR_LP_GEMM_HALF=$h timeout -k 10 200 python3 bench.py --model C --batch $b --precision $prec > $O/o.out 2> $O/o.err
  echo "batch $b $prec half=$h rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6))")" | tee -a $O/summary.txt
done; done; done
