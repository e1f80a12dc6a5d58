#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_order; mkdir -p $O
for c in fp8 bf16_fp8 f32_fp8 f32_bf16_fp8 fp8_withA bf16_withA; do
  timeout -k 10 300 python3 tools/experiments/chain_order_check.py $c 2>&1 | grep "inf/s" | tee -a $O/summary.txt
done
