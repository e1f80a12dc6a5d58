#!/bin/bash
# round 5: value_pcie_inclusive inside bench.py (other legs' streams alive) -- run-to-run spread, and the H2D on the worker's own stream (FR_HOST_ZEROCOPY=2) for comparison
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
P=$R/gpu-fpga-recommendation-system_amd
for rep in 1 2 3; do
  for zc in 3 2; do
    FR_LIB=$P/libfleetrec_exp.so FR_HOST_ZEROCOPY=$zc python3 $R/bench.py --legs pcie 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('zc=$zc value %.2f M  pcie_inclusive %.2f M  ratio %.4f' % (j['value']/1e6, j['value_pcie_inclusive']/1e6, j['value_pcie_inclusive']/j['value']))"
  done
done 2>&1 | tee $R/gpurun_out/r05_pcie_var.txt
