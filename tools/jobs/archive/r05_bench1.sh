#!/bin/bash
# round 5: the driver's own command + the 4-rank share-device test
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench1.json 2> gpurun_out/r05_bench1.err
echo "rc=$?"; tail -3 gpurun_out/r05_bench1.err; cut -c1-3000 gpurun_out/r05_bench1.json
cp gpurun_out/bench_detail.json gpurun_out/r05_bench1_detail.json
timeout -k 10 900 python -m pytest tests/test_dist_gloo.py -x -q -m gpu -k "four_ranks" 2>&1 | tail -15
