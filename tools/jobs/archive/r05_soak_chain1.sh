#!/bin/bash
# round 5: race screen of a lone worker's chain (chain width 1: fc_pp_gemm_n128_kernel)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(timeout -k 10 300 python3 $R/tools/soak_chain.py 90 bf16 1 2>&1 | tail -3 && timeout -k 10 300 python3 $R/tools/soak_chain.py 90 fp8 1 2>&1 | tail -3) | tee $R/gpurun_out/r05_soak_chain1.txt
