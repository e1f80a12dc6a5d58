#!/bin/bash
# host-fed streaming (index rows in host memory, scores back to host memory): host threads x workers per thread
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_pcie; mkdir -p $O
for td in "4 2" "4 3" "4 4" "6 2" "8 1" "8 2"; do set -- $td
  timeout -k 10 200 python3 bench.py --legs pcie --threads $1 --depth $2 > $O/o.out 2> $O/o.err
  echo "threads=$1 depth=$2 rc=$? $(python3 -c "
import json
d=json.load(open('gpurun_out/bench_detail.json'))
print('value %.2f M  pcie_inclusive_streaming %.2f M  per-batch %.2f M' % (d['value']/1e6, d['pcie_inclusive_streaming']['value']/1e6, d['pcie_inclusive']['value']/1e6))")" | tee -a $O/summary.txt
done
