#!/bin/bash
# the next step's DMA instructions issued between this step's MFMAs (IL) against the burst behind the barrier: parity, FC1 side by side, the chain
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_il; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
FR_LP_GEMM_IL=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "half_chip or gemm_256 or tiled_gemm_model_c or how_many_workers" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc $(tail -1 $O/pytest.log)"
[ $rc -ne 0 ] && { tail -30 $O/pytest.log; exit 1; }
for rep in 1 2; do for prec in bf16 fp8; do for il in 0 1; do
  FR_LP_GEMM_IL=$il timeout -k 10 200 python3 bench.py --model C --batch 4096 --precision $prec > $O/o.out 2> $O/o.err
  echo "$prec IL=$il rc=$? $(python3 -c "
import json
d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); rf=d['roofline']
print('%.2f M  FC1 %s %.1f us frac %.3f' % (d['value']/1e6, rf['kernel_name'], 1e3*rf['avg_launch_ms'], rf['frac']))")" | tee -a $O/summary.txt
done; done; done
