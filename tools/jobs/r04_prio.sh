#!/bin/bash
# worker streams spread over stream priorities (one hardware-queue pool per priority) against GPU_MAX_HW_QUEUES=8 and the default
set -o pipefail
O=gpurun_out/r04_prio; mkdir -p $O
export FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
run() { local label=$1; shift
  timeout -k 10 200 env "$@" > $O/o.out 2> $O/o.err
  echo "$label rc=$? $(python3 -c "import json,sys; d=json.loads(open('$O/o.out').read().strip().splitlines()[-1]); print('%.2f M' % (d['value']/1e6))")" | tee -a $O/summary.txt
}
for args in "--model C --batch 4096 --precision bf16" "--model C --batch 4096 --precision fp8" "--model B --batch 1024 --precision bf16" "--model A --batch 256 --precision f32"; do
  run "$args default" FR_STREAM_PRIO=0 python3 bench.py $args
  run "$args prio2" FR_STREAM_PRIO=1 python3 bench.py $args
  run "$args prio3" FR_STREAM_PRIO=2 python3 bench.py $args
  run "$args hwq8" GPU_MAX_HW_QUEUES=8 python3 bench.py $args
done
