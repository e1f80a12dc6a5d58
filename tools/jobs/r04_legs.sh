#!/bin/bash
# why is Model-C fp8 slower on the default line than alone?  the configs legs after different sets of earlier legs
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_legs; mkdir -p $O
for legs in configs "cpu,configs" "groups,configs" "pcie,configs" "tcp,configs" "roofline,configs"; do
  timeout -k 10 400 python3 bench.py --legs $legs > $O/o.out 2> $O/o.err
  echo "legs=$legs rc=$? $(python3 -c "
import json
d=json.load(open('gpurun_out/bench_detail.json'))
print(' '.join('%s %.1f' % (c['tag'], c['value']/1e6) for c in d['configs'] if c['tag'].startswith('C4096')))")" | tee -a $O/summary.txt
done
