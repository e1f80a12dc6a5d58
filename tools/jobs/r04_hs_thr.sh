#!/bin/bash
# where the persistent bf16 kernel starts to pay now that it carries no scratch: tiles per compute unit 1, 1.25, 1.5, 2 (Model-A batch 256: 64 tiles per
# 64 batches... group g = g/64 tiles per CU; Model-B batch 1024: group 16 = one tile per CU)
set -o pipefail
O=gpurun_out/r04_hs_thr; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for cfg in "A 256 64" "A 256 80" "A 256 96" "A 256 128" "B 1024 16" "B 1024 20" "B 1024 24" "B 1024 32"; do set -- $cfg
  for hk in 0 1; do
    FR_FUSED_HK=$hk timeout -k 10 200 python3 bench.py --model $1 --batch $2 --precision bf16 --group $3 > $O/o.out 2> $O/o.err
    echo "$1 batch $2 group $3 hk=$hk rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); r=d['roofline']; print('%.2f M  %s %.1f us' % (d['value']/1e6, r['kernel_name'], 1e3*r['avg_launch_ms']))")" | tee -a $O/summary.txt
  done
done
