#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu -k "staged_exchange or c_abi_step" > gpurun_out/r06_staged1.log 2>&1
rc=$?
tail -25 gpurun_out/r06_staged1.log
exit $rc
