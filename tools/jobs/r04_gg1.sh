#!/bin/bash
set -o pipefail
O=gpurun_out/r04_gg1; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "gather_inside_fc1" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -15 $O/pytest.log
[ $rc -ne 0 ] && exit 1
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for prec in bf16 fp8; do for pb in "" "--per-bank"; do for gg in 0 1; do
  FR_GEMM_GATHER=$gg timeout -k 10 200 python3 bench.py --model C --batch 4096 --precision $prec $pb > $O/o.out 2> $O/o.err
  echo "$prec $pb gemm_gather=$gg rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6), [round(1e3*x,1) for x in d.get('layer_launch_ms')])")" | tee -a $O/summary.txt
done; done; done
