#!/bin/bash
# round 5: which unit the phased-waves GEMM tile keeps busy (tools/pmc_gemm.sh: separate --pmc passes; the counter passes serialise the launches: a launch alone on half the chip -- double the per-256-CU fractions)
cd $GRAFT_REPO_ROOT
bash tools/pmc_gemm.sh bf16 r05_pp_bf16 > gpurun_out/r05_pmc_gemm_bf16.txt 2>&1; tail -3 gpurun_out/r05_pmc_gemm_bf16.txt | cut -c1-1500
bash tools/pmc_gemm.sh fp8 r05_pp_fp8 > gpurun_out/r05_pmc_gemm_fp8.txt 2>&1; tail -3 gpurun_out/r05_pmc_gemm_fp8.txt | cut -c1-1500
cp gpurun_out/pmc_gemm/r05_pp_bf16.json gpurun_out/r05_pmc_gemm_pp_bf16.json; cp gpurun_out/pmc_gemm/r05_pp_fp8.json gpurun_out/r05_pmc_gemm_pp_fp8.json
rm -rf gpurun_out/pmc_gemm
