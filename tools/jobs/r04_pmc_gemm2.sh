#!/bin/bash
# CU-side counters of the part-chip GEMM kernels (four workers on the context; counter passes serialise the launches: each runs ALONE on its part of the chip)
cd $GRAFT_REPO_ROOT
for prec in bf16 fp8; do
  bash tools/pmc_gemm.sh $prec r04_C4096_${prec}_part_chip > gpurun_out/pmc_gemm2_$prec.log 2>&1; tail -5 gpurun_out/pmc_gemm2_$prec.log | cut -c1-1500
done
mkdir -p gpurun_out/r04_pmc_gemm2; cp gpurun_out/pmc_gemm/*.json gpurun_out/r04_pmc_gemm2/; rm -rf gpurun_out/pmc_gemm
