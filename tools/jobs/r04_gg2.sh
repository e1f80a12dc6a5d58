#!/bin/bash
# kernel durations of the Model-C chain with the gather inside the FC1 launch (experiments build, FR_GEMM_GATHER=1) and without
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_gg2; mkdir -p $O
export FR_LIB=$R/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
cd /tmp && export TMPDIR=/tmp
for prec in bf16 fp8; do for gg in 0 1; do for td in "1 1" "2 2"; do set -- $td
  (cd $R && FR_GEMM_GATHER=$gg timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 bench.py --model C --batch 4096 --precision $prec --threads $1 --depth $2 --quick > $O/o.out 2> $O/o.err)
  f=$(ls $O/t/*/*kernel_stats.csv | head -1)
  echo "== $prec gemm_gather=$gg threads=$1 depth=$2  value $(python3 -c "import json; print('%.2f M' % (json.loads(open('$O/o.out').read().strip().splitlines()[-1])['value']/1e6))")" | tee -a $O/summary.txt
  python3 - $f <<'PY' | tee -a $O/summary.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("gemm", "gather_out", "fr_pipeline_kernel<4")):
        print("   %-60s calls %5s avg %8.1f us" % (n.split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $O/t
done; done; done
