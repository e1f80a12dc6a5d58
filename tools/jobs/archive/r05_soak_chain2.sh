#!/bin/bash
# round 5: race screen at batch 1024, four workers (FC1 on fc_pp_gemm_n128_kernel beside its neighbours)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(timeout -k 10 300 python3 $R/tools/soak_chain.py 60 bf16 4 1024 2>&1 | tail -3 && timeout -k 10 300 python3 $R/tools/soak_chain.py 60 fp8 4 1024 2>&1 | tail -3) | tee $R/gpurun_out/r05_soak_chain2.txt
