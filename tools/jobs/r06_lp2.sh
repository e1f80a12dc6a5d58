#!/bin/bash
# round 6: kernel-level view of the operand-type bank image A/B (rocprofv3 --kernel-trace --stats per configuration)
set -o pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06_lp
for prec in bf16 fp8; do for on in 0 1; do for mode in lone chains; do
  tag=${prec}_img${on}_${mode}
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r06_lp/$tag -o t -- python3 $R/tools/experiments/lp_image_trace.py $prec $on $mode > $R/gpurun_out/r06_lp/$tag.log 2>&1 || { tail -5 $R/gpurun_out/r06_lp/$tag.log; exit 1; }
  f=$(find $R/gpurun_out/r06_lp/$tag -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/r06_lp/${tag}_kernel_stats.csv
  rm -rf $R/gpurun_out/r06_lp/$tag
  echo "== $tag"; head -6 $R/gpurun_out/r06_lp/${tag}_kernel_stats.csv | cut -c1-200
done; done; done
