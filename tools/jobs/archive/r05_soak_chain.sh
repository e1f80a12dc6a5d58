#!/bin/bash
# round 5: race screen of the Model-C chain on the phased-waves GEMM tile: four workers, every batch's scores bit for bit against a lone worker's
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(timeout -k 10 400 python3 $R/tools/soak_chain.py 150 bf16 2>&1 | tail -4 && timeout -k 10 400 python3 $R/tools/soak_chain.py 150 fp8 2>&1 | tail -4) | tee $R/gpurun_out/r05_soak_chain.txt
