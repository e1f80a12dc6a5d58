#!/bin/bash
# Model-C rows of the default line (legs=configs, 2 x 2) under the stream forms of the experiments build
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_legs3; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for sp in -1 0 3; do
  FR_STREAM_PRIO=$sp timeout -k 10 400 python3 bench.py --legs configs > $O/o.out 2> $O/o.err
  echo "default line, legs=configs, STREAM_PRIO=$sp rc=$? $(python3 -c "
import json
d=json.load(open('gpurun_out/bench_detail.json'))
print(' '.join('%s %.1f' % (c['tag'], c['value']/1e6) for c in d['configs'] if c['tag'].startswith('C')))")" | tee -a $O/summary.txt
done
