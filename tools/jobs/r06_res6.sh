#!/bin/bash
mkdir -p gpurun_out
L=$PWD/gpu-fpga-recommendation-system_amd
FR_LIB=$L/libfleetrec_exp.so SETTLE=1 timeout -k 10 400 python tools/experiments/queue_aging.py > gpurun_out/r06_queue_aging_settle.txt 2>&1
tail -20 gpurun_out/r06_queue_aging_settle.txt
