#!/bin/bash
# two minutes of the Model-C chain per low precision (driver re-created every fifth window), then one minute of the headline
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_soak; mkdir -p $O
for prec in bf16 fp8; do
  timeout -k 10 300 python3 tools/experiments/soak_chain.py $prec 120 > $O/$prec.txt 2>&1; echo "$prec rc=$? $(tail -1 $O/$prec.txt)"
  python3 - $O/$prec.txt <<'PY'
import re, sys
v = [float(x) for x in re.findall(r"([0-9.]+) M inferences/s", open(sys.argv[1]).read())]
print("   windows %d: min %.2f  median %.2f  max %.2f M inf/s" % (len(v), min(v), sorted(v)[len(v)//2], max(v)))
PY
done
timeout -k 10 200 python3 tools/experiments/soak.py 60 > $O/headline.txt 2>&1; echo "headline rc=$?"; python3 - $O/headline.txt <<'PY'
import re, sys
v = [float(x) for x in re.findall(r"([0-9.]+) M inferences/s", open(sys.argv[1]).read())]
print("   windows %d: min %.2f  median %.2f  max %.2f M inf/s" % (len(v), min(v), sorted(v)[len(v)//2], max(v)))
PY
