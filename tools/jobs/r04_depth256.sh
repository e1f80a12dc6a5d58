#!/bin/bash
# 256 x 256 tile with 4-row steps and 4 / 5 stages in LDS (more loads in flight) against the shipped 2 x 8 rows
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_depth256; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for deep in 0 4 5; do
  FR_LP_GEMM_256_DEPTH=$deep timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "half_chip or gemm_256" > $O/pytest_$deep.log 2>&1; rc=$?; echo "depth $deep pytest rc=$rc $(tail -1 $O/pytest_$deep.log)"
  [ $rc -ne 0 ] && { tail -30 $O/pytest_$deep.log; exit 1; }
  for prec in bf16 fp8; do
    FR_LP_GEMM_256_DEPTH=$deep timeout -k 10 200 python3 bench.py --model C --batch 4096 --precision $prec > $O/o.out 2> $O/o.err
    echo "$prec depth=$deep rc=$? $(python3 -c "
import json
d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); rf=d['roofline']
print('%.2f M  FC1 %s %.1f us conc %.2f frac %.3f' % (d['value']/1e6, rf['kernel_name'], 1e3*rf['avg_launch_ms'], rf.get('concurrent_launches',0), rf['frac']))")" | tee -a $O/summary.txt
  done
done
