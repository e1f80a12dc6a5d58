#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python tools/experiments/residency_candidates.py > gpurun_out/r06_residency_candidates.txt 2>&1
tail -30 gpurun_out/r06_residency_candidates.txt
