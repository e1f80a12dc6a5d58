#!/bin/bash
# which hardware queue every worker stream's kernels run on: Model-C fp8 chain in a fresh process against after a closed Model-A driver
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_queues; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in fp8 preA_fp8; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/$c -o t -- python3 $R/tools/experiments/chain_order_check.py $c > $O/$c.out 2>&1
  grep "inf/s" $O/$c.out
  f=$(find $O/$c -name "*kernel_trace.csv" | head -1)
  python3 - $f <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print("columns:", [k for k in rows[0].keys() if "Id" in k or "id" in k])
m = collections.Counter()
for r in rows[-20000:]:
    n = r["Kernel_Name"]
    if "gemm" in n or "gather_out" in n:
        m[(r.get("Stream_Id"), r.get("Queue_Id"))] += 1
for k, v in sorted(m.items()): print("  stream %s queue %s: %d kernels" % (k[0], k[1], v))
PY
  rm -rf $O/$c
done
