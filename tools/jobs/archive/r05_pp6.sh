#!/bin/bash
# round 5: fc_pp_gemm_kernel vs fc_lp_gemm_kernel on ONE box, with and without rocprofv3 (bf16 FC1, two launches side by side, --roofline-only)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05_pp6
mkdir -p $O
export FR_LIB=$R/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
cd /tmp && export TMPDIR=/tmp
show() { python3 -c "
import json,sys
for ln in open(sys.argv[1]):
    if ln.startswith('{') and '\"metric\"' in ln:
        j=json.loads(ln); print('   events: layers ms %s conc %s %s' % ([round(x,4) for x in j['layer_launch_ms']], [round(x,2) for x in j['layer_concurrency']], j['layer_kernels'][:1]))" $1; }
for rep in 1 2; do
for prec in bf16 fp8; do
for pp in 0 3 2; do
  export FR_LP_GEMM_PP=$pp
  echo "== $prec FR_LP_GEMM_PP=$pp unprofiled"
  timeout -k 10 300 python3 $R/bench.py --roofline-only --model C --batch 4096 --precision $prec > $O/u.log 2>/dev/null; show $O/u.log
  echo "== $prec FR_LP_GEMM_PP=$pp under rocprofv3 --kernel-trace --stats"
  rm -rf $O/st; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st -- python3 $R/bench.py --roofline-only --model C --batch 4096 --precision $prec > $O/p.log 2>&1; show $O/p.log
  f=$(ls $O/st/*/*kernel_stats.csv | head -1); grep "gemm_kernel<[12], [23]" $f | head -2 | cut -c1-160
done
done
done 2>&1 | tee $O/../r05_pp_profiler_ab.txt
rm -rf $O
