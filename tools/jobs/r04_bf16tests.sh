#!/bin/bash
set -o pipefail
O=gpurun_out/r04_bf16tests; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16 or persistent or custom or streaming or fused or group" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $O/pytest.log
