#!/bin/bash
# record gather, per-table indices: rows of tables above a size threshold fetched with nt loads (experiments build, FR_GATHER_NT_KB)
set -o pipefail
O=gpurun_out/r04_nt1; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for kb in 0 64 256 1024 4096 32768 262144; do
  FR_GATHER_NT_KB=$kb timeout -k 10 200 python3 bench.py --legs gather,bank --no-gather-ab --quick > $O/o.out 2> $O/o.err
  echo "nt_kb=$kb rc=$? $(python3 -c "
import json
d=json.load(open('gpurun_out/bench_detail.json')); g=d['gather']; z=g.get('zipf_1.05',{}); b=d.get('gather_per_bank',{})
print('uniform %.1f us frac %.3f | zipf %.1f us frac %.3f | per-bank %.1f us frac %.3f' % (1e3*g['avg_launch_ms'], g['frac'], 1e3*z.get('avg_launch_ms',0), z.get('frac',0), 1e3*b.get('avg_launch_ms',0), b.get('frac',0)))")" | tee -a $O/summary.txt
done
