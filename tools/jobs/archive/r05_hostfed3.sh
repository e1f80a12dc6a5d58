#!/bin/bash
# round 5: host-fed stream: host threads x workers at one stream per hardware queue, with / without the D2H copy command
set -o pipefail
mkdir -p gpurun_out/r05_hostfed
R=$GRAFT_REPO_ROOT
export FR_LIB=$R/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for zc in 0 2; do
  for cfg in "4 1" "2 2" "1 4" "3 1" "2 1" "4 2" "8 1"; do
    echo "== FR_HOST_ZEROCOPY=$zc threads x depth = $cfg"
    FR_HOST_ZEROCOPY=$zc python3 $R/tools/host_fed_run.py $cfg 1.0 2>&1 | grep -v "^[EW]2"
  done
done 2>&1 | tee $R/gpurun_out/r05_hostfed/threads_ab.txt
