#!/bin/bash
# round 6: operand-type bank image -- parity test, then the per-bank Model-C chain legs with the image on / off
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "operand_type_bank_image or per_bank or chain_width" > gpurun_out/r06_lp1_tests.log 2>&1
rc=$?
tail -15 gpurun_out/r06_lp1_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/experiments/lp_image_ab.py > gpurun_out/r06_lp_image_ab.txt 2>&1
rc=$?
cat gpurun_out/r06_lp_image_ab.txt | tail -30
exit $rc
