#!/bin/bash
# round 6: soak of the host-fed streaming path (workers created / destroyed along the way: the context reference counts) and of the headline's sustained rate
mkdir -p gpurun_out
timeout -k 10 300 python tools/soak_host_fed.py 120 4 2>&1 | tail -4 | tee gpurun_out/r06_soak.txt
timeout -k 10 300 python tools/experiments/soak.py 90 2>&1 | tail -6 | tee -a gpurun_out/r06_soak.txt
