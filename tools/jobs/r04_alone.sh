#!/bin/bash
# every kernel of the Model-C chain ALONE on the chip: one worker, one stream (rocprofv3 averages)
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_alone; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for prec in bf16 fp8; do for pb in "" "--per-bank"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/bench.py --model C --batch 4096 --precision $prec $pb --threads 1 --depth 1 --quick > $O/o.log 2>&1 || echo failed
  f=$(ls $O/st/*/*kernel_stats.csv | head -1)
  echo "== $prec $pb: $(grep -o '"value": [0-9.e+]*' $O/o.log | tail -1)" | tee -a $O/summary.txt
  python3 - $f <<'PY' | tee -a $O/summary.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) > 50: print("   %-60s calls %5s avg %7.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
  rm -rf $O/st
done; done
