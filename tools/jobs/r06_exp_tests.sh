#!/bin/bash
# round 6: the experiments-library-only tests (skipped in the product suite) after this round's changes
mkdir -p gpurun_out
L=$PWD/gpu-fpga-recommendation-system_amd/libfleetrec_exp.so
FR_LIB=$L FR_GEMM_GATHER=1 timeout -k 10 400 python -m pytest tests/test_gpu_lowprec.py -x -q -m gpu -k "streaming_gather_inside_fc1" 2>&1 | tail -3
FR_LIB=$L FR_FUSED_HK=1 timeout -k 10 400 python -m pytest tests/test_gpu_lowprec.py -x -q -m gpu -k "fp8_persistent" 2>&1 | tail -3
FR_LIB=$L FR_FUSED_LP_ROWS=1 timeout -k 10 400 python -m pytest tests/test_gpu_lowprec.py -x -q -m gpu -k "operand_type_rows" 2>&1 | tail -3
FR_LIB=$L timeout -k 10 400 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu -k "sharded_fc_failure" 2>&1 | tail -3
