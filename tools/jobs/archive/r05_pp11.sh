#!/bin/bash
# round 5: fc_pp_gemm_kernel with the bf16 accumulators pinned in AccVGPRs (libfleetrec_ppagpr.so, `make ppagpr`) against the experiments build: parity, then A/B
set -o pipefail
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
P=$R/gpu-fpga-recommendation-system_amd
FR_LIB=$P/libfleetrec_exp.so timeout -k 10 300 python3 $R/tools/experiments/gemm_pp_check.py /tmp/ppv.npz 2>&1 | tail -1 || exit 1
FR_LIB=$P/libfleetrec_ppagpr.so timeout -k 10 300 python3 $R/tools/experiments/gemm_pp_check.py /tmp/ppa.npz 2>&1 | tail -1 || exit 1
timeout 60 python3 $R/tools/experiments/gemm_pp_check.py /tmp/ppv.npz /tmp/ppa.npz 2>&1 | grep -v "^kernel" | tee $R/gpurun_out/r05_pp_agpr.txt
for rep in 1 2 3; do
  for lib in exp ppagpr; do
    for bank in "" "--per-bank"; do
    echo "== bf16 $bank lib=$lib"
    FR_LIB=$P/libfleetrec_$lib.so timeout -k 10 300 python3 $R/bench.py --model C --batch 4096 --precision bf16 $bank 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('   %.2f M inf/s  layers ms %s  conc %s' % (j['value']/1e6, [round(x,4) for x in j['layer_launch_ms']], [round(x,2) for x in j.get('layer_concurrency')]))" || exit 1
    done
  done
done 2>&1 | tee -a $R/gpurun_out/r05_pp_agpr.txt
