#!/bin/bash
# round 5: one-rank rehearsal of the sharded step loop under a kernel trace: how much of a step is kernels, how much the host loop
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_sharded_trace; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp && export TMPDIR=/tmp
for ex in alltoall; do
for prec in bf16; do
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -o t -- python3 $R/bench.py --gpus 1 --backend nccl --force-process-group --mode sharded --precision $prec --exchange $ex --steps 400 --warmup 20 > $O/out.json 2> $O/err.txt
grep -o '"ms_per_step": [0-9.]*' $O/out.json | head -2
f=$(find $O/t -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_overlap.py $f 2>&1 | head -60 | tee $O/overlap.txt
rm -rf $O/t
done
done
