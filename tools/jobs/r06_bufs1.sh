#!/bin/bash
# does the headline's rotation of 64 index buffers (770 k distinct rows = 98 MB of lines: inside the 256 MB Infinity Cache) flatter `value`?
mkdir -p gpurun_out
for n in 64 1024 4096 64 4096; do
  sed "s/^N_IDX_BUFFERS = 64 /N_IDX_BUFFERS = $n /" bench.py > /tmp/bench_n.py
  cp /tmp/bench_n.py bench_n_tmp.py
  timeout -k 10 200 python bench_n_tmp.py --legs roofline 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('index buffers $n: value %.2f M inf/s; kernel %.1f us per launch, frac %.4f' % (j['value']/1e6, 1e3*j['roofline']['avg_launch_ms'], j['roofline']['frac']))" | tee -a gpurun_out/r06_headline_index_buffers.txt
  rm -f bench_n_tmp.py
done
