#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python bench.py > gpurun_out/r06_bench2.json 2> gpurun_out/r06_bench2.err
rc=$?
python3 - <<'PY'
import json
j = json.loads(open("gpurun_out/r06_bench2.json").read().strip().splitlines()[-1])
print(j["value"], j["roofline"]["frac"])
for k, v in j["other_configs"].items(): print(k, v)
PY
tail -3 gpurun_out/r06_bench2.err
exit $rc
