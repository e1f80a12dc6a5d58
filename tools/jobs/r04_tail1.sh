#!/bin/bash
set -o pipefail
O=gpurun_out/r04_tail1; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for prec in bf16 fp8; do for pb in "" "--per-bank"; do for tail in 0 1; do
  FR_FC_TAIL=$tail timeout -k 10 200 python3 bench.py --model C --batch 4096 --precision $prec $pb > $O/o.out 2> $O/o.err
  echo "$prec $pb tail=$tail rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6), d.get('layer_launch_ms'))")" | tee -a $O/summary.txt
done; done; done
