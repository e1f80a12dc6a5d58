#!/bin/bash
# round 5: the sharded path after the one-launch slice transpose (no whole-image memset) and chain width 1: tests, then the one-rank rehearsal + its trace
set -o pipefail
R=$GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 900 python -m pytest $R/tests -x -q -m gpu -k "shard or rccl or slices or config5 or cfg" 2>&1 | tail -4 || exit 1
bash $R/tools/jobs/r05_sharded_trace.sh 2>&1 | head -40
for ex in allgather alltoall; do
for prec in f32 bf16 fp8; do
timeout -k 10 300 python3 $R/bench.py --gpus 1 --backend nccl --force-process-group --mode sharded --precision $prec --exchange $ex --steps 400 --warmup 20 2> /tmp/sh.err | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); c=j['config']
print('sharded $prec $ex: %.1f M inf/s, %.1f us per step, pipelined==stepwise %s, vs unsharded %s' % (j['value']/1e6, 1e3*j['ms_per_step'], c['pipelined_equals_stepwise'], c['sharded_vs_unsharded_context']))" || tail -5 /tmp/sh.err
done
done
