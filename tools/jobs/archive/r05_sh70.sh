#!/bin/bash
set -o pipefail
timeout -k 10 800 python -m pytest $GRAFT_REPO_ROOT/tests/test_gpu_parity.py -x -q -m gpu -k "single_device_emulation" --durations=5 2>&1 | tail -12
