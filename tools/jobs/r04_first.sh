#!/bin/bash
# round 4, first GPU call: the whole GPU suite, then the driver's own bench command (the compact line)
set -o pipefail
mkdir -p gpurun_out/r04_first
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_first/pytest.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r04_first/pytest.log
tail -5 gpurun_out/r04_first/pytest.log
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_first/bench.out 2> gpurun_out/r04_first/bench.err
echo "bench rc=$?"
tail -c 4200 gpurun_out/r04_first/bench.out
cp gpurun_out/bench_detail.json gpurun_out/r04_first/bench_detail.json
