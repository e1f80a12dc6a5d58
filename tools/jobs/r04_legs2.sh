#!/bin/bash
# Model-C rows of the default line: driver threads 4 x 1 against 2 x 2; and the single-configuration run for reference
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_legs2; mkdir -p $O
for td in "2 2" "4 1"; do set -- $td
  timeout -k 10 400 python3 bench.py --legs configs --threads $1 --depth $2 > $O/o.out 2> $O/o.err
  echo "default line, legs=configs, $1x$2 rc=$? $(python3 -c "
import json
d=json.load(open('gpurun_out/bench_detail.json'))
print(' '.join('%s %.1f' % (c['tag'], c['value']/1e6) for c in d['configs']))")" | tee -a $O/summary.txt
done
timeout -k 10 200 python3 bench.py --model C --batch 4096 --precision fp8 > $O/o.out 2> $O/o.err
echo "single fp8: $(python3 -c "
import json
d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6))")" | tee -a $O/summary.txt
