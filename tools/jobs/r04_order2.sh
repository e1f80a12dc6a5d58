#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_order2; mkdir -p $O
for c in fp8 preA_fp8 preB_fp8 preA_all_fp8 preB_all_fp8 preA_preB_all_fp8; do
  timeout -k 10 300 python3 tools/experiments/chain_order_check.py $c 2>&1 | grep "inf/s" | tee -a $O/summary.txt
done
