#!/bin/bash
# round 5: soak of the host-fed streaming path (copy stream + pinned scores) and of the CPU back-end's pool under concurrent callers
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout -k 10 200 python3 $R/tools/soak_host_fed.py 60 4 2>&1 | tail -2 | tee $R/gpurun_out/r05_soak.txt
