#!/bin/bash
# 256 x 256 bf16 GEMM tile on v_mfma_f32_16x16x32_bf16 (MF16) against 32x32x16: parity, FC1 side by side, the chain
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_mf16; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
FR_LP_GEMM_MF16=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "half_chip or gemm_256 or tiled_gemm_model_c" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc $(tail -1 $O/pytest.log)"
[ $rc -ne 0 ] && { tail -30 $O/pytest.log; exit 1; }
for rep in 1 2; do for mf in 0 1; do for pb in "" "--per-bank"; do
  FR_LP_GEMM_MF16=$mf timeout -k 10 200 python3 bench.py --model C --batch 4096 --precision bf16 $pb > $O/o.out 2> $O/o.err
  echo "bf16 $pb MF16=$mf rc=$? $(python3 -c "
import json
d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); rf=d['roofline']
print('%.2f M  FC1 %s %.1f us conc %.2f frac %.3f' % (d['value']/1e6, rf['kernel_name'], 1e3*rf['avg_launch_ms'], rf.get('concurrent_launches',0), rf['frac']))")" | tee -a $O/summary.txt
done; done; done
for mf in 0 1; do
  FR_LP_GEMM_MF16=$mf timeout -k 10 200 python3 bench.py --model C --batch 8192 --precision bf16 > $O/o.out 2> $O/o.err
  echo "bf16 batch 8192 MF16=$mf rc=$? $(python3 -c "
import json
d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6))")" | tee -a $O/summary.txt
done
