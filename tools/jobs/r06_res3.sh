#!/bin/bash
mkdir -p gpurun_out
L=$PWD/gpu-fpga-recommendation-system_amd
echo "== product library (queue probe on)" > gpurun_out/r06_queue_aging2.txt
FR_LIB=$L/libfleetrec.so timeout -k 10 400 python tools/experiments/queue_aging.py >> gpurun_out/r06_queue_aging2.txt 2>&1
echo "== experiments library, FR_QUEUE_PROBE=0 (round 5's behaviour)" >> gpurun_out/r06_queue_aging2.txt
FR_LIB=$L/libfleetrec_exp.so FR_QUEUE_PROBE=0 timeout -k 10 400 python tools/experiments/queue_aging.py >> gpurun_out/r06_queue_aging2.txt 2>&1
tail -24 gpurun_out/r06_queue_aging2.txt
