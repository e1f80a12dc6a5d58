#!/bin/bash
# round 5: host-fed stream: H2D on a copy stream ahead of the launches + scores written to pinned memory (FR_HOST_ZEROCOPY=3, the new default) against round 4's commands (0)
set -o pipefail
mkdir -p gpurun_out/r05_hostfed
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest $R/tests/test_gpu_parity.py -x -q -m gpu -k "host_fed or test_random_streaming_sequences" 2>&1 | tail -3
export FR_LIB=$R/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for rep in 1 2; do
for zc in 0 3 2; do
  for cfg in "4 2" "2 2" "4 1" "8 2"; do
    echo "== FR_HOST_ZEROCOPY=$zc threads x depth = $cfg"
    FR_HOST_ZEROCOPY=$zc python3 $R/tools/host_fed_run.py $cfg 1.0 2>&1 | grep -v "^[EW]2"
  done
done
done 2>&1 | tee $R/gpurun_out/r05_hostfed/copy_stream_ab.txt
