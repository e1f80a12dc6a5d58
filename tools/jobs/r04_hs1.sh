#!/bin/bash
set -o pipefail
O=gpurun_out/r04_hs1; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "persistent or bf16 or fused or streaming or committed_fc" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2; do
for cfg in "B 1024 bf16" "B 1024 bf16 --per-bank" "A 256 bf16"; do
  timeout -k 10 200 python3 bench.py --model ${cfg%% *} --batch $(echo $cfg | cut -d' ' -f2) --precision bf16 $(echo $cfg | cut -d' ' -f4-) > $O/o.out 2> $O/o.err
  echo "$cfg rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); r=d['roofline']; print('%.2f M  kernel %s %.1f us frac %.3f' % (d['value']/1e6, r['kernel_name'], 1e3*r['avg_launch_ms'], r['frac']))")" | tee -a $O/summary.txt
done; done
