#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest $R/tests/test_gpu_parity.py -x -q -m gpu -k "phased_waves or gemm_256" 2>&1 | tail -25
