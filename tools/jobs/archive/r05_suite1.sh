#!/bin/bash
# round 5: the whole GPU suite after the chain-width / host-fed / CPU back-end changes
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu --durations=12 2>&1 | tee gpurun_out/r05_suite1.log | tail -40
