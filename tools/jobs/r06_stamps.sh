#!/bin/bash
# round 6: who arrives last at FC1's slice barriers -- the product's producers (fp32 rows, two row sets) vs operand-type rows with four row sets (diagnostic build)
mkdir -p gpurun_out
L=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_diag.so
echo "== fp32 rows, D = 2 (the product's instantiation, diagnostic build)" > gpurun_out/r06_hs_stamps.txt
FR_LIB=$L timeout -k 10 200 python tools/experiments/fused_hk_stamps.py 128 B >> gpurun_out/r06_hs_stamps.txt 2>&1
echo "== bf16 rows, D = 4 (FR_FUSED_LP_ROWS=1)" >> gpurun_out/r06_hs_stamps.txt
FR_LIB=$L FR_FUSED_LP_ROWS=1 timeout -k 10 200 python tools/experiments/fused_hk_stamps.py 128 B >> gpurun_out/r06_hs_stamps.txt 2>&1
tail -60 gpurun_out/r06_hs_stamps.txt
