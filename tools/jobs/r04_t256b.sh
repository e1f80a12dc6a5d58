#!/bin/bash
set -o pipefail
O=gpurun_out/r04_t256b; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "gemm_256_tile or tiled_gemm" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
