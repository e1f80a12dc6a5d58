#!/bin/bash
# host-fed streaming with the workers on their own hardware queues (experiments build: FR_STREAM_PRIO=1 forces the priority alternation for fused-kernel models too)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_pcie2; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for sp in -1 1 -1 1; do for td in "2 2"; do set -- $td
  FR_STREAM_PRIO=$sp timeout -k 10 200 python3 bench.py --legs pcie,tcp --threads $1 --depth $2 > $O/o.out 2> $O/o.err
  echo "STREAM_PRIO=$sp threads=$1 depth=$2 rc=$? $(python3 -c "
import json
d=json.load(open('gpurun_out/bench_detail.json'))
print('value %.2f M  pcie_inclusive_streaming %.2f M  per-batch %.2f M  tcp %.2f M' % (d['value']/1e6, d['pcie_inclusive_streaming']['value']/1e6, d['pcie_inclusive']['value']/1e6, (d.get('tcp_streaming') or {}).get('value', 0)/1e6))")" | tee -a $O/summary.txt
done; done
