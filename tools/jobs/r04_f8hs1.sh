#!/bin/bash
set -o pipefail
O=gpurun_out/r04_f8hs1; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
FR_FUSED_HK=1 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "fp8_persistent" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest.log
[ $rc -ne 0 ] && exit 1
for sched in chunked 63 50 42 84; do
  if [ $sched = chunked ]; then export FR_FUSED_HK=0; else export FR_FUSED_HK=1 FR_FUSED_F8_SCHED=$sched; fi
  timeout -k 10 200 python3 bench.py --model B --batch 1024 --precision fp8 --group 128 > $O/o.out 2> $O/o.err
  echo "B fp8 sched=$sched rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); r=d['roofline']; print('%.2f M  kernel %s %.1f us frac %.3f' % (d['value']/1e6, r['kernel_name'], 1e3*r['avg_launch_ms'], r['frac']))")" | tee -a $O/summary.txt
done
for hk in 0 1; do for g in 64 256; do
  FR_FUSED_HK=$hk timeout -k 10 200 python3 bench.py --model A --batch 256 --precision fp8 --group $g > $O/o.out 2> $O/o.err
  echo "A fp8 hk=$hk group=$g rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); r=d['roofline']; print('%.2f M  kernel %s %.1f us frac %.3f' % (d['value']/1e6, r['kernel_name'], 1e3*r['avg_launch_ms'], r['frac']))")" | tee -a $O/summary.txt
done; done
