#!/bin/bash
set -o pipefail
O=gpurun_out/r04_custom; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "custom or persistent" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
