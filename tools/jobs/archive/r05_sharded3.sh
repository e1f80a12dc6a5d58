#!/bin/bash
# round 5: the host side of the sharded step loop: one-rank rehearsal at small batches (the GPU work shrinks, the host loop does not)
set -o pipefail
R=$GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
for b in 512 1024 4096; do
for ex in allgather alltoall; do
timeout -k 10 300 python3 $R/bench.py --gpus 1 --backend nccl --force-process-group --mode sharded --precision bf16 --exchange $ex --batch $b --steps 1000 --warmup 50 2> /tmp/sh.err | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); c=j['config']
print('sharded bf16 $ex batch $b: %.1f M inf/s, %.1f us per step' % (j['value']/1e6, 1e3*j['ms_per_step']))" || tail -5 /tmp/sh.err
done
done
