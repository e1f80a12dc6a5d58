#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu --durations=8 > gpurun_out/r06_suite2.log 2>&1
rc=$?
tail -18 gpurun_out/r06_suite2.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python tools/experiments/chain_concurrency_margin.py > gpurun_out/r06_chain_concurrency_margin.txt 2>&1
tail -8 gpurun_out/r06_chain_concurrency_margin.txt
FR_LIB=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so FR_FUSED_LP_ROWS=1 timeout -k 10 300 python -m pytest tests/test_gpu_lowprec.py -x -q -m gpu -k "operand_type_rows" 2>&1 | tail -3
