#!/bin/bash
# does the number of HIP hardware queues change the 4-stream chains?  (the kernel trace shows 4 worker streams on 2 hardware queues)
set -o pipefail
O=gpurun_out/r04_hwq; mkdir -p $O
for q in default 8; do
  for cfg in "C 4096 bf16" "C 4096 fp8" "B 1024 bf16" "A 256 f32"; do
    set -- $cfg
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    if [ "$1" = A ]; then
      timeout -k 10 200 python3 bench.py --legs none --steps 2000 --warmup 500 > $O/${q}_$1_$3.out 2> $O/${q}_$1_$3.err
    else
      timeout -k 10 200 python3 bench.py --model $1 --batch $2 --precision $3 > $O/${q}_$1_$3.out 2> $O/${q}_$1_$3.err
    fi
    echo "queues=$q $cfg rc=$? value=$(python3 -c "import json,sys; print('%.2f M' % (json.loads(open('$O/${q}_$1_$3.out').read().strip().splitlines()[-1])['value']/1e6))")"
  done
done
