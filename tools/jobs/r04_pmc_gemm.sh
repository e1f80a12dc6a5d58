#!/bin/bash
# CU-side counters of the Model-C GEMM kernels (FC1 bf16 / fp8, and the narrow FC2 / FC3 kernel): which unit is busy
cd $GRAFT_REPO_ROOT
for prec in bf16 fp8; do
  bash tools/pmc_gemm.sh $prec r04_C4096_$prec > gpurun_out/pmc_gemm_$prec.log 2>&1; tail -4 gpurun_out/pmc_gemm_$prec.log | cut -c1-1800
done
mkdir -p gpurun_out/r04_pmc_gemm; cp gpurun_out/pmc_gemm/*.json gpurun_out/r04_pmc_gemm/; rm -rf gpurun_out/pmc_gemm
