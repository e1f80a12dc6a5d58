#!/bin/bash
set -o pipefail
O=gpurun_out/r04_pair1; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "batches_in_pairs or tiled_gemm_model_c or streaming or gemm_256" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -12 $O/pytest.log
[ $rc -ne 0 ] && exit 1
for prec in bf16 fp8; do for pb in "" "--per-bank"; do for g in 0 2; do
  timeout -k 10 200 python3 bench.py --model C --batch 4096 --precision $prec $pb --group $g > $O/o.out 2> $O/o.err
  echo "$prec $pb group=$g rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6))")" | tee -a $O/summary.txt
done; done; done
