#!/bin/bash
# kernel timeline of the Model-C batch-4096 chain (4 streams): which kernels overlap, how long each runs beside the others
set -o pipefail
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r04_trace_c2; mkdir -p $O
cd $GRAFT_REPO_ROOT
for prec in bf16 fp8; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/$prec -o t -- python3 bench.py --model C --batch 4096 --precision $prec --quick > $O/$prec.out 2> $O/$prec.err
  echo "$prec rc=$?"; tail -c 600 $O/$prec.out
  f=$(find $O/$prec -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_overlap.py $f > $O/${prec}_overlap.txt 2>&1
  tail -40 $O/${prec}_overlap.txt
  # keep only the tail of the raw trace (steady state), the box returns at most 64 MiB
  tail -n 6000 $f > $O/${prec}_trace_tail.csv; head -1 $f > $O/${prec}_trace_head.csv
  rm -rf $O/$prec
done
