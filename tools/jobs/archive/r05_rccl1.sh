#!/bin/bash
# round 5: the RCCL code path of the N > 1 bench line rehearsed on ONE GPU: a one-rank torch.distributed process group on the nccl (= RCCL) backend,
# every collective of the line (max / sum over ranks on device tensors, the sharded legs' all-gather on RCCL's stream with the external-stream hand-over)
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 600 python3 $R/bench.py --gpus 1 --backend nccl --force-process-group --legs roofline --quick --rows-cap 200000 > $R/gpurun_out/r05_rccl_line.json 2> $R/gpurun_out/r05_rccl_line.err
echo "default line rc=$?"; tail -1 $R/gpurun_out/r05_rccl_line.err; echo "stdout lines: $(wc -l < $R/gpurun_out/r05_rccl_line.json)"; python3 -c "
import json
j=json.loads(open('$R/gpurun_out/r05_rccl_line.json').read().splitlines()[-1])
print({k: j[k] for k in ('value','gather_per_bank_all_ranks','configs_all_ranks','sharded') if k in j})"
for ex in allgather alltoall; do
for prec in bf16 fp8; do
timeout -k 10 300 python3 $R/bench.py --gpus 1 --backend nccl --force-process-group --mode sharded --precision $prec --exchange $ex --steps 20 --warmup 5 2> $R/gpurun_out/r05_rccl_sh.err | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); c=j['config']
print('sharded $prec $ex: %.1f M inf/s, pipelined==stepwise %s, vs unsharded %s, backend %s' % (j['value']/1e6, c['pipelined_equals_stepwise'], c['sharded_vs_unsharded_context'], c['backend']))" || tail -5 $R/gpurun_out/r05_rccl_sh.err
done
done
