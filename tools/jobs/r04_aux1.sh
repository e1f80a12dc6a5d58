#!/bin/bash
# aux-stream gather (Model-C chain): parity tests of the chain, then throughput A/B (experiments build: FR_GATHER_AUX=0/1), then a trace
set -o pipefail
O=gpurun_out/r04_aux1; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "model_c or committed_fc or tiled_gemm or streaming or chain or sharded or properties" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for aux in 0 1; do for prec in bf16 fp8; do for pb in "" "--per-bank"; do
  FR_GATHER_AUX=$aux timeout -k 10 200 python3 bench.py --model C --batch 4096 --precision $prec $pb > $O/aux${aux}_${prec}${pb}.out 2> $O/aux${aux}_${prec}${pb}.err
  echo "aux=$aux $prec $pb rc=$? value=$(python3 -c "import json,sys; print('%.2f M' % (json.loads(open('$O/aux${aux}_${prec}${pb}.out').read().strip().splitlines()[-1])['value']/1e6))")"
done; done; done
unset FR_LIB
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for prec in bf16 fp8; do
  (cd $R && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_$prec -o t -- python3 bench.py --model C --batch 4096 --precision $prec --quick > $R/$O/tr_$prec.out 2> $R/$O/tr_$prec.err)
  f=$(find $R/$O/tr_$prec -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/trace_overlap.py $f > $R/$O/${prec}_overlap.txt 2>&1
  head -14 $R/$O/${prec}_overlap.txt; tail -24 $R/$O/${prec}_overlap.txt
  cp $f $R/$O/${prec}_trace.csv; rm -rf $R/$O/tr_$prec
done
