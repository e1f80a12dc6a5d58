#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python tools/experiments/queue_aging.py > gpurun_out/r06_queue_aging_pool.txt 2>&1
tail -12 gpurun_out/r06_queue_aging_pool.txt
