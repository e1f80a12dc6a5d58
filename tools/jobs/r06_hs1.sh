#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "operand_type or bf16_persistent or custom_model_rides or bf16_chain" > gpurun_out/r06_hs1_tests.log 2>&1
rc=$?
tail -12 gpurun_out/r06_hs1_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/experiments/lp_rows_hs_ab.py > gpurun_out/r06_lp_rows_hs_ab.txt 2>&1
rc=$?
tail -12 gpurun_out/r06_lp_rows_hs_ab.txt
exit $rc
