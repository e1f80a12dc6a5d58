#!/bin/bash
# per-layer chip time of the Model-C chain, four workers side by side (bench.py --roofline-only), product library and full-chip tiles
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_layers; mkdir -p $O
for prec in bf16 fp8; do for part in -1 1; do for pb in "" "--per-bank"; do
  FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so FR_LP_GEMM_PART=$part timeout -k 10 200 python3 bench.py --roofline-only --model C --batch 4096 --precision $prec $pb > $O/o.out 2> $O/o.err
  echo "$prec part=$part $pb rc=$? $(python3 -c "
import json
d=json.loads(open('$O/o.out').read().strip().splitlines()[-1])
print(' | '.join('%s %.1f us / %.2f = %.1f' % (k.split('<')[1][:14], 1e3*m, c, 1e3*m/c) for k,m,c in zip(d['layer_kernels'], d['layer_launch_ms'], d['layer_concurrency'])))")" | tee -a $O/summary.txt
done; done; done
