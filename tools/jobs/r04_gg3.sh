#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_gg3; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "gather_inside_fc1" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest.log
[ $rc -ne 0 ] && exit 1
export FR_LIB=$R/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for prec in bf16 fp8; do for pb in "" "--per-bank"; do for gg in 0 1; do
  FR_GEMM_GATHER=$gg timeout -k 10 200 python3 bench.py --model C --batch 4096 --precision $prec $pb > $O/o.out 2> $O/o.err
  echo "$prec $pb gemm_gather=$gg rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6))")" | tee -a $O/summary.txt
done; done; done
cd /tmp && export TMPDIR=/tmp
for prec in bf16 fp8; do for td in "1 1" "2 2"; do set -- $td
  (cd $R && FR_GEMM_GATHER=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 bench.py --model C --batch 4096 --precision $prec --threads $1 --depth $2 --quick > $O/o.out 2> $O/o.err)
  f=$(ls $O/t/*/*kernel_stats.csv | head -1)
  echo "== $prec gemm_gather=1 threads=$1 depth=$2  value $(python3 -c "import json; print('%.2f M' % (json.loads(open('$O/o.out').read().strip().splitlines()[-1])['value']/1e6))")" | tee -a $O/summary.txt
  grep -E "gemm_gather|fc_lp_gemm_kernel<., 1, 64|pipe_kernel<1" $f | awk -F'","' '{printf "   %-50s calls %s avg %.1f us\n", substr($1,2,50), $2, $4/1e3}' | tee -a $O/summary.txt
  rm -rf $O/t
done; done
