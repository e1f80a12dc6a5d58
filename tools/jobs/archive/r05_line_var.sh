#!/bin/bash
# round 5: the driver's command three times on one box: spread of value / value_pcie_inclusive / the roofline fraction
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for rep in 1 2 3; do
  timeout -k 10 600 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
oc=j['other_configs']
print('value %.2f M  pcie %.2f M (%.4f)  roofline %.4f  gather_per_bank %.4f  B1024_bf16 %.1f M  C4096 bf16 %.2f M fp8 %.2f M  per bank %.2f / %.2f M' % (j['value']/1e6, j['value_pcie_inclusive']/1e6, j['value_pcie_inclusive']/j['value'], j['roofline']['frac'], j['gather_per_bank']['frac'], oc['B1024_bf16']['inf_per_s']/1e6, oc['C4096_bf16']['inf_per_s']/1e6, oc['C4096_fp8']['inf_per_s']/1e6, oc['C4096_bf16_per_bank']['inf_per_s']/1e6, oc['C4096_fp8_per_bank']['inf_per_s']/1e6))"
done 2>&1 | tee $R/gpurun_out/r05_line_var.txt
