#!/bin/bash
# round 5: kernel timeline of the Model-C batch-4096 chain (2 x 2 workers) on the final GEMM kernels: which kernels overlap, how long each runs beside the others
set -o pipefail
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_trace_c; mkdir -p $O
cd $GRAFT_REPO_ROOT
for prec in bf16 fp8; do
 for bank in "" "--per-bank"; do
  n=$prec${bank:+_per_bank}
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/$n -o t -- python3 bench.py --model C --batch 4096 --precision $prec $bank --throughput-only > $O/$n.out 2> $O/$n.err
  echo "$n rc=$?"; grep -o '"value": [0-9.]*' $O/$n.out | head -1
  f=$(find $O/$n -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_overlap.py $f > $O/${n}_overlap.txt 2>&1
  head -48 $O/${n}_overlap.txt
  rm -rf $O/$n
 done
done
