#!/bin/bash
# round 5: value_pcie_inclusive with the copy streams in the lowest-priority queue pool: six runs of the driver's command's first legs on one box
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout -k 10 300 python -m pytest $R/tests/test_gpu_parity.py -x -q -m gpu -k "host_fed or random_streaming" 2>&1 | tail -2 || exit 1
for rep in 1 2 3 4 5 6; do
  timeout -k 10 600 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --legs pcie,roofline,groups 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
print('value %.2f M  pcie %.2f M (%.4f)' % (j['value']/1e6, j['value_pcie_inclusive']/1e6, j['value_pcie_inclusive']/j['value']))"
done 2>&1 | tee $R/gpurun_out/r05_pcie_var2.txt
