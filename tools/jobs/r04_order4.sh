#!/bin/bash
# chain workers on the highest / lowest stream priority only: fresh process, after closed contexts, beside an idle one; then the default line's rows
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_order4; mkdir -p $O
for c in fp8 preA_fp8 preA_preB_all_fp8 preA_bf16 fp8_withA; do
  timeout -k 10 300 python3 tools/experiments/chain_order_check.py $c 2>&1 | grep "inf/s" | grep -v ": pre" | tee -a $O/summary.txt
done
timeout -k 10 400 python3 bench.py --legs configs,bank --no-gather-ab > $O/o.out 2> $O/o.err
echo "default line, legs=configs,bank rc=$? $(python3 -c "
import json
d=json.load(open('gpurun_out/bench_detail.json'))
print(' '.join('%s %.1f' % (c['tag'], c['value']/1e6) for c in d['configs'] if c['tag'].startswith('C')))")" | tee -a $O/summary.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr -o t -- python3 $R/tools/experiments/chain_order_check.py preA_fp8 > $O/tr.out 2>&1
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY' | tee -a $O/summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
m = collections.Counter()
for r in rows[-20000:]:
    n = r["Kernel_Name"]
    if "gemm" in n or "gather_out" in n:
        m[(r.get("Stream_Id"), r.get("Queue_Id"))] += 1
for k, v in sorted(m.items()): print("  preA_fp8 under the profiler: stream %s queue %s: %d kernels" % (k[0], k[1], v))
PY
rm -rf $O/tr
