#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_dist_gloo.py -x -q -m gpu -k "cabi_rccl or two_ranks or four_ranks" 2>&1 | tail -5
