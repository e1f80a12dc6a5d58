#!/bin/bash
set -o pipefail
O=gpurun_out/r04_aux3; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for prec in bf16 fp8; do for td in "1 1" "1 2" "2 2"; do for aux in "0 0" "1 0" "2 0" "2 1" "1 1"; do
  set -- $td $aux
  FR_GATHER_AUX=$3 FR_GATHER_AUX_PRIO=$4 timeout -k 10 200 python3 bench.py --model C --batch 4096 --precision $prec --threads $1 --depth $2 > $O/o.out 2> $O/o.err
  echo "$prec threads=$1 depth=$2 aux=$3 prio=$4 rc=$? value=$(python3 -c "import json,sys; print('%.2f M' % (json.loads(open('$O/o.out').read().strip().splitlines()[-1])['value']/1e6))")" | tee -a $O/summary.txt
done; done; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(cd $R && FR_GATHER_AUX=2 FR_GATHER_AUX_PRIO=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr -o t -- python3 bench.py --model C --batch 4096 --precision bf16 --threads 1 --depth 1 --quick > $R/$O/tr.out 2> $R/$O/tr.err)
f=$(find $R/$O/tr -name "*kernel_trace.csv" | head -1)
python3 $R/tools/trace_overlap.py $f > $R/$O/overlap.txt 2>&1
head -12 $R/$O/overlap.txt; tail -24 $R/$O/overlap.txt
rm -rf $R/$O/tr
