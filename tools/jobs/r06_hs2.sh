#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for d in 2 4; do
  echo "== FR_FUSED_LP_D=$d (experiments build)"
  FR_FUSED_LP_D=$d timeout -k 10 300 python tools/experiments/lp_rows_hs_ab.py 2>&1 | tee -a gpurun_out/r06_lp_rows_hs_ab2.txt | tail -8
done
