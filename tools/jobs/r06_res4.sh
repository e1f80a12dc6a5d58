#!/bin/bash
mkdir -p gpurun_out
L=$PWD/gpu-fpga-recommendation-system_amd
FR_LIB=$L/libfleetrec_exp.so FR_QUEUE_PROBE=0 timeout -k 10 400 python tools/experiments/queue_aging.py > gpurun_out/r06_queue_aging3.txt 2>&1
tail -40 gpurun_out/r06_queue_aging3.txt
