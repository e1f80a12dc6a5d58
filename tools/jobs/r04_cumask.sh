#!/bin/bash
# own hardware queue per worker through a full CU mask (equal priority) against the priority spread and plain streams
set -o pipefail
cd $GRAFT_REPO_ROOT; O=gpurun_out/r04_cumask; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
for prec in bf16 fp8 f32; do for sp in 0 2 3; do
  FR_STREAM_PRIO=$sp timeout -k 10 200 python3 bench.py --model C --batch 4096 --precision $prec > $O/o.out 2> $O/o.err
  echo "$prec STREAM_PRIO=$sp rc=$? $(python3 -c "
import json
d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); det=d
rf=d['roofline']; print('%.2f M  FC1 %.1f us conc %.2f frac %.3f  busy %s' % (d['value']/1e6, 1e3*rf['avg_launch_ms'], rf.get('concurrent_launches',0), rf['frac'], (det.get('layer_stream_busy_ms') or [None])[0]))")" | tee -a $O/summary.txt
done; done
