#!/bin/bash
# round 6: the C-ABI RCCL leg of the N > 1 bench line -- (1) its child alone with a one-rank communicator, (2) the parent's integration and failure
# path with two ranks sharing the one GPU (RCCL refuses two ranks on one device: the line must carry the error and exit 0)
mkdir -p gpurun_out
ID=$(python -c "
import __graft_entry__ as g
fr=g.load_package(); print(fr.Comm.unique_id().hex())" 2>/dev/null | tail -1)
timeout -k 10 200 python bench.py --cabi-child 0/1/0/$ID/gpurun_out/r06_cabi_child.json; echo "child rc $?"; cat gpurun_out/r06_cabi_child.json; echo
FR_BENCH_CABI_RCCL=force timeout -k 10 600 python bench.py --gpus 2 --backend gloo --share-device --legs none --steps 300 --warmup 100 > gpurun_out/r06_cabi_line.json 2> gpurun_out/r06_cabi_line.err; echo "line rc $?"
python3 -c "
import json
j=json.loads(open('gpurun_out/r06_cabi_line.json').read().strip().splitlines()[-1])
print('n_gpus', j['n_gpus'], 'value', j['value'], 'sharded ok', (j.get('sharded') or {}).get('ok'), 'cabi', j.get('sharded_cabi_rccl'))"
tail -3 gpurun_out/r06_cabi_line.err
