#!/bin/bash
# round 6: race screen of the chain over the operand-type bank image (references computed on fp32 rows), bf16 and fp8, four workers
mkdir -p gpurun_out
for prec in bf16 fp8; do
  timeout -k 10 200 python tools/soak_chain.py 75 $prec 4 4096 bank 2>&1 | tail -3 | tee -a gpurun_out/r06_soak_chain_bank.txt
done
