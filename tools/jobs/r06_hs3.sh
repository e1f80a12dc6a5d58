#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=$PWD/gpu-fpga-recommendation-system_amd
for mode in table bank; do
  FR_LIB=$L/libfleetrec.so timeout -k 10 120 python tools/experiments/hs_ablate_rate.py $mode 2>&1 | tail -1 | tee -a gpurun_out/r06_hs_ablate.txt
  for ab in 0 1 2; do
    FR_LIB=$L/libfleetrec_diag.so FR_FUSED_HS_ABLATE=$ab timeout -k 10 120 python tools/experiments/hs_ablate_rate.py $mode 2>&1 | tail -1 | tee -a gpurun_out/r06_hs_ablate.txt
  done
done
