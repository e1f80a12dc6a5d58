#!/bin/bash
# round 5, job 1: the configs[3] / configs[4] single-GPU emulation tests at full size (VERDICT r04 item 1 a, b)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "test_table_sharded_full_size_g8 or test_config5_all_shards_fp8_chain" --durations=5 2>&1 | tee gpurun_out/r05_cfg34.log
