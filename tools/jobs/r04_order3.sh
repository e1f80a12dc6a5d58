#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_order3; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so FR_DRIVER_TIMING=1
for c in fp8 preA_fp8; do
  timeout -k 10 300 python3 tools/experiments/chain_order_check.py $c 2>&1 | grep "inf/s\|driver thread" | tail -8 | tee -a $O/summary.txt
done
