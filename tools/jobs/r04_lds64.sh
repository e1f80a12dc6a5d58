#!/bin/bash
# 256 x 256 tile with 4-row steps: 64 / 96 / 128 KiB of LDS per workgroup -- does a smaller GEMM workgroup start sooner beside the gather's workgroups?
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_lds64; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
FR_LP_GEMM_256_ROWS=4 FR_LP_GEMM_256_STAGES=2 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "half_chip or gemm_256" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc $(tail -1 $O/pytest.log)"
[ $rc -ne 0 ] && { tail -30 $O/pytest.log; exit 1; }
for prec in bf16 fp8; do for cfg in "8 2" "4 2" "4 3" "4 4"; do set -- $cfg
  for pb in "" "--per-bank"; do
  FR_LP_GEMM_256_ROWS=$1 FR_LP_GEMM_256_STAGES=$2 timeout -k 10 200 python3 bench.py --model C --batch 4096 --precision $prec $pb > $O/o.out 2> $O/o.err
  echo "$prec $pb rows=$1 stages=$2 rc=$? $(python3 -c "
import json
d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); rf=d['roofline']
print('%.2f M  FC1 %s %.1f us frac %.3f' % (d['value']/1e6, rf['kernel_name'], 1e3*rf['avg_launch_ms'], rf['frac']))")" | tee -a $O/summary.txt
  done
done; done
