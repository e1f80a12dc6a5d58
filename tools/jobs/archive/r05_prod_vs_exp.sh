#!/bin/bash
# round 5: does the product build run any leg slower than the experiments build (whose runtime knobs change what hipcc may schedule)?  Same box, the default line of each.
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for rep in 1 2; do
for lib in libfleetrec.so libfleetrec_exp.so; do
  FR_LIB=$R/gpu-fpga-recommendation-system_amd/$lib timeout -k 10 600 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
l=json.loads(sys.stdin.read().splitlines()[-1])
oc=l['other_configs']
print('$lib', 'value %.2f M pcie %.2f M roof %.3f bank %.3f gather %.3f |' % (l['value']/1e6, l['value_pcie_inclusive']/1e6, l['roofline']['frac'], l['gather_per_bank']['frac'], l['gather']['frac']), ' '.join('%s %.1f' % (k, v['inf_per_s']/1e6) for k,v in oc.items()))" || exit 1
done
done 2>&1 | tee $R/gpurun_out/r05_prod_vs_exp.txt
